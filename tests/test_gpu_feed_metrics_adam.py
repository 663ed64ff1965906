"""GPU: the callers either side of the encoder path, moved to the device.

  f1  ops.assemble_batch / train.DeviceFeed vs the reference-recorded DatasetTrain stream (index_selection.json):
      bit-exact history / candidate / label / mask for every sample of every shard.
  f2  ops.eval_metrics vs the reference-recorded metric fixture (metrics.json, <= 1e-6 as the task states; measured
      ~1e-15) and vs the package's numpy metrics on 10^5 random impressions incl. ties, single-class impressions and
      long candidate lists.
  f4  parallel.FlatBucket (HIP nr_adam_step, gradients accumulated in place) vs torch.optim.Adam on the same model and
      batches: same parameters after 10 steps (<= 2e-6 abs, fp32 rounding of one fused update)."""
import argparse
import json
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from newsrecommendation_amd import data as D, metrics as M, ops, parallel, train as TR

pytestmark = pytest.mark.gpu


def test_device_batch_assembly_is_bit_exact_with_the_reference_stream(tmp_path):
    g = json.load(open(os.path.join(GOLDEN, "index_selection.json")))
    args = argparse.Namespace(user_log_length=g["user_log_length"], npratio=g["npratio"])
    rnd = np.random.RandomState(0)
    comb = rnd.randint(1, 1000, size=(len(g["news_index"]) + 1, 7)).astype(np.int32)      # [N+1, F] feature rows
    comb[0] = 0
    for n_shards, case in g["cases"].items():
        for r in range(int(n_shards)):
            f = tmp_path / f"train_{n_shards}_{r}.tsv"
            f.write_text("".join(case["train_shards"][r]))
            stream = case["train_stream"][r]                   # [hist idx, mask, sample idx, label] per line, from the reference
            shard = D.IndexedTrainShard(str(f), g["news_index"], args)
            feed = TR.DeviceFeed(shard, comb, batch_size=4, device="cuda")
            random.seed(g["seed"] + r)                         # the seed the fixture's stream was recorded under
            feed.start_epoch()
            got_h, got_m, got_c, got_l = [], [], [], []
            for i in range(len(feed)):
                h, m, c, l = feed.batch(i)
                got_h.append(h.cpu()); got_m.append(m.cpu()); got_c.append(c.cpu()); got_l.append(l.cpu())
            h, m, c, l = (torch.cat(x).numpy() for x in (got_h, got_m, got_c, got_l))
            assert h.shape[0] == len(stream)
            for i, (hist, mask, sample, label) in enumerate(stream):
                assert np.array_equal(h[i], comb[hist]) and np.array_equal(c[i], comb[sample])
                assert m[i].tolist() == mask and int(l[i]) == label
            # the very same tensors the host DataLoader path yields
            random.seed(g["seed"] + r)
            ds = D.DatasetTrain(str(f), g["news_index"], comb, args)
            hb, mb, cb, lb = next(iter(torch.utils.data.DataLoader(ds, batch_size=len(stream))))
            assert np.array_equal(hb.numpy(), h) and np.array_equal(cb.numpy(), c) and np.array_equal(lb.numpy(), l)


def test_assemble_batch_reports_bad_indices():
    comb = torch.arange(12, dtype=torch.int32).reshape(4, 3).cuda()
    hist = torch.tensor([[0, 3]], dtype=torch.int32).cuda()
    pos, neg = torch.tensor([1], dtype=torch.int32).cuda(), torch.tensor([[2, 9]], dtype=torch.int32).cuda()
    ops.CHECK_INDICES = True
    try:
        with pytest.raises(IndexError):
            ops.assemble_batch(comb, hist, pos, neg, torch.tensor([0]).cuda())
        h, c = ops.assemble_batch(comb, hist, pos, torch.tensor([[2, 3]], dtype=torch.int32).cuda(), torch.tensor([2]).cuda())
        assert c[0].tolist() == [[6, 7, 8], [9, 10, 11], [3, 4, 5]] and h[0].tolist() == [[0, 1, 2], [9, 10, 11]]
    finally:
        ops.CHECK_INDICES = False


def _host_metrics(labels, scores):
    out, n = np.zeros(4), 0
    for y, s in zip(labels, scores):
        if y.mean() in (0, 1):
            continue
        out += [M.roc_auc_score(y, s), M.mrr_score(y, s), M.ndcg_score(y, s, 5), M.ndcg_score(y, s, 10)]
        n += 1
    return n, out


def _csr(labels, scores):
    off = np.zeros(len(labels) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(y) for y in labels])
    return (torch.from_numpy(np.concatenate(scores).astype(np.float32)).cuda(), torch.from_numpy(np.concatenate(labels).astype(np.int32)).cuda(),
            torch.from_numpy(off).cuda())


def test_eval_metrics_against_the_reference_fixture():
    recs = json.load(open(os.path.join(GOLDEN, "metrics.json")))
    labels = [np.array(r["y"]) for r in recs]
    scores = [np.array(r["s"], dtype=np.float32) for r in recs]
    sums, per = ops.eval_metrics(*_csr(labels, scores), return_per_impression=True)
    per = per.cpu().numpy()
    for i, r in enumerate(recs):
        assert np.allclose(per[i], r["auc_mrr_ndcg5_ndcg10"], atol=1e-6), (i, per[i], r["auc_mrr_ndcg5_ndcg10"])
        assert np.abs(per[i] - np.array(r["auc_mrr_ndcg5_ndcg10"])).max() < 1e-12
    want = np.sum([r["auc_mrr_ndcg5_ndcg10"] for r in recs], axis=0)
    assert int(sums[0]) == len(recs) and np.allclose(sums[1:].cpu().numpy(), want, atol=1e-9)


def test_eval_metrics_on_1e5_random_impressions_with_ties_and_skips():
    rnd = np.random.RandomState(1)
    n = 100000
    labels, scores = [], []
    for i in range(n):
        c = int(rnd.randint(2, 101)) if i % 1000 else int(rnd.randint(300, 2000))      # mean ~51, a few long lists
        y = (rnd.rand(c) < 0.15).astype(np.int64)
        if i % 97 == 0:
            y[:] = 0                                                                  # skipped (src/main.py:250)
        if i % 89 == 0:
            y[:] = 1
        s = rnd.randn(c).astype(np.float32)
        if i % 5 == 0:
            s = np.round(s * 2) / 2                                                   # heavy ties
        labels.append(y); scores.append(s)
    sums, per = ops.eval_metrics(*_csr(labels, scores), return_per_impression=True)
    sums2 = ops.eval_metrics(*_csr(labels, scores))
    assert torch.equal(sums, sums2)                                                   # fixed-order reduction: bit-reproducible
    per = per.cpu().numpy()
    tie_free = [i for i in range(n) if i % 5 and labels[i].mean() not in (0, 1)]
    for i in tie_free[:3000]:
        y, s = labels[i], scores[i]
        want = [M.roc_auc_score(y, s), M.mrr_score(y, s), M.ndcg_score(y, s, 5), M.ndcg_score(y, s, 10)]
        assert np.abs(per[i] - want).max() < 1e-12, (i, per[i], want)
    # ties: AUC (average ranks) is order independent and must agree exactly; MRR / nDCG use the documented tie order
    # (stable ascending sort reversed)
    for i in [i for i in range(0, n, 5) if labels[i].mean() not in (0, 1)][:1500]:
        y, s = labels[i], scores[i]
        order = np.argsort(s, kind="stable")[::-1]
        yt = y[order]
        mrr = np.sum(yt / (np.arange(len(yt)) + 1)) / yt.sum()
        dcg = lambda k: np.sum((2 ** yt[:k] - 1) / np.log2(np.arange(len(yt[:k])) + 2))
        want = [M.roc_auc_score(y, s), mrr, dcg(5) / M.dcg_score(y, y, 5), dcg(10) / M.dcg_score(y, y, 10)]
        assert np.abs(per[i] - want).max() < 1e-12, (i, per[i], want)
    skipped = sum(1 for y in labels if y.mean() in (0, 1))
    assert int(sums[0]) == n - skipped and (per[:, 0] < 0).sum() == skipped
    keep = per[:, 0] >= 0
    assert np.allclose(sums[1:].cpu().numpy(), per[keep].sum(0), rtol=1e-12)


def _tiny_nrms(dt="fp32", freeze=False):
    from oracle import nr_oracle as O
    from newsrecommendation_amd.model import NRMS
    cfg = O.default_cfg(num_words_title=8, user_log_length=6, npratio=2, word_embedding_dim=32, news_dim=32, num_attention_heads=4,
                        news_query_vector_dim=16, user_query_vector_dim=16, drop_rate=0.0, freeze_embedding=freeze)
    g = torch.Generator().manual_seed(0)
    table = torch.randn(200, 32, generator=g) * 0.4
    table[0] = 0
    torch.manual_seed(5)
    m = NRMS.Model(SimpleNamespace(**vars(cfg), compute_dtype=dt), table.numpy()).cuda().train()
    batches = []
    for _ in range(10):
        hist = torch.randint(0, 200, (16, 6, 8), generator=g, dtype=torch.int32).cuda()
        cand = torch.randint(1, 200, (16, 3, 8), generator=g, dtype=torch.int32).cuda()
        mask = (torch.rand(16, 6, generator=g) < 0.7).float().cuda()
        label = torch.randint(0, 3, (16,), generator=g).cuda()
        batches.append((hist, mask, cand, label))
    return m, batches


def test_adam_kernel_matches_torch_adam_on_the_same_gradients():
    """nr_adam_step alone: the same gradient stream into torch.optim.Adam and into the HIP kernel (with the 1/world factor
    folded in as grad_scale and the zero_grad folded in), odd length (scalar tail)."""
    from newsrecommendation_amd import _lib
    n = 100003
    g = torch.Generator(device="cuda").manual_seed(0)
    p0 = torch.randn(n, device="cuda", generator=g)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=3e-4)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 13):
        grad = torch.randn(n, device="cuda", generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,))))
        pt.grad = grad * 0.5
        opt.step()
        gbuf = grad.clone()
        _lib.check(_lib.lib().nr_adam_step(p.data_ptr(), gbuf.data_ptr(), m.data_ptr(), v.data_ptr(), n, 3e-4, 0.9, 0.999, 1e-8, step, 0.5, 1,
                                           torch.cuda.current_stream().cuda_stream), "nr_adam_step")
        assert float(gbuf.abs().max()) == 0.0
    st = opt.state[pt]
    assert float((p - pt.detach()).abs().max()) <= 1e-6
    # the moments carry gradients of very different magnitudes (1e-6 .. 1): one fp32 ulp of the largest contribution
    assert float((m - st["exp_avg"]).abs().max()) <= 2e-5 * float(st["exp_avg"].abs().max())
    assert float((v - st["exp_avg_sq"]).abs().max()) <= 2e-5 * float(st["exp_avg_sq"].abs().max())


@pytest.mark.parametrize("freeze", [False, True])
def test_flat_bucket_fused_adam_matches_torch_adam(freeze):
    m1, batches = _tiny_nrms(freeze=freeze)
    m2, _ = _tiny_nrms(freeze=freeze)
    assert all(torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m2.state_dict().values()))
    opt = torch.optim.Adam(m1.parameters(), lr=1e-3)
    fb = parallel.FlatBucket(m2, lr=1e-3)
    l1, l2 = [], []
    for b in batches:
        loss, _ = m1(*b)
        opt.zero_grad()
        loss.backward()
        opt.step()
        l1.append(float(loss))
        loss, _ = m2(*b)
        loss.backward()                     # straight into fb.grad (no autograd accumulation, no zero fill)
        fb.step()
        l2.append(float(loss))
    assert max(abs(a - b) for a, b in zip(l1, l2)) < 2e-5, (l1, l2)
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert n1 == n2
        if n1.endswith(("W_K.bias", "att_fc2.bias")):
            # analytically zero gradients (a constant shift of all keys / all pooling logits changes nothing): what is left
            # is rounding noise, which Adam normalises to steps of +-lr -- not comparable between two runs of ANY optimizer
            continue
        assert float((p1 - p2).abs().max()) <= 5e-5, n1      # 10 Adam steps of lr 1e-3: parameters moved by ~1e-2
    assert float(fb.grad.abs().max()) == 0.0                # zero_grad folded into the kernel
    # the moments, exposed in torch.optim.Adam's layout
    st = fb.state_dict()["state"]
    for n, p in m1.named_parameters():
        if p.requires_grad:
            assert float((opt.state[p]["exp_avg"] - st[n]["exp_avg"]).abs().max()) <= 1e-6 + 1e-3 * float(opt.state[p]["exp_avg"].abs().max()), n


def test_flat_bucket_checkpoint_round_trip(tmp_path):
    """src/main.py:118-142 layout, written from a FlatBucket model (parameters are views of one buffer) and read back with
    the loader that executes nothing from the file; keys / shapes as the reference's state_dict (Appendix A)."""
    m, batches = _tiny_nrms()
    fb = parallel.FlatBucket(m, lr=1e-3)
    loss, _ = m(*batches[0])
    loss.backward()
    fb.step()
    path = os.path.join(tmp_path, "epoch-1.pt")
    torch.save(TR.checkpoint_dict(m, {"news": 1}, {"sub": 2}), path)
    assert os.path.getsize(path) < 4 * sum(p.numel() for p in m.parameters()) * 1.5 + 65536      # no whole-bucket copies per view
    ck = TR.load_checkpoint(path)
    assert ck["category_dict"] == {"news": 1} and ck["subcategory_dict"] == {"sub": 2}
    m2, _ = _tiny_nrms()
    m2.load_state_dict(ck["model_state_dict"], strict=True)
    m.eval(); m2.eval()
    with torch.no_grad():
        a, b = m(*batches[1]), m2(*batches[1])
    assert torch.equal(a[1], b[1])


def test_pack_cache_follows_parameter_updates():
    """ops.pack_cache: the packed (bf16, padded / transposed) copies of weights that live in a FlatBucket are served from a
    cache and refreshed -- all of them in one nr_cast_pad_batch launch -- when the fused Adam kernel (ops.param_epoch) or a
    torch-side in-place write (tensor version) changed the parameters.  Tensors outside a bucket are never cached."""
    from newsrecommendation_amd import ops, parallel
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(400, 200), torch.nn.Linear(200, 1200)).cuda()
    fb = parallel.FlatBucket(net, lr=1e-2)
    ws = [net[0].weight, net[1].weight]

    def expect(w, transpose):
        ref = (w.detach().t() if transpose else w.detach()).to(torch.bfloat16)
        return ref

    def check_all():
        outs = []
        for w in ws:
            for tr in (False, True):
                pk = ops.pack(w, ops.NR_BF16, transpose=tr)
                ref = expect(w, tr)
                assert torch.equal(pk[:, :ref.shape[1]], ref)
                assert float(pk[:, ref.shape[1]:].abs().sum()) == 0.0          # zero padded
                outs.append(pk)
        return outs

    a = check_all()
    b = check_all()
    assert all(x.data_ptr() == y.data_ptr() for x, y in zip(a, b))               # second round: served from the cache
    fb.grad.fill_(0.5)
    fb.adam_step()                                                               # parameters change behind autograd's back
    check_all()
    with torch.no_grad():
        ws[0].mul_(2.0)                                                          # torch-side in-place write
    check_all()
    loose = torch.randn(64, 96, device="cuda")
    p1, p2 = ops.pack(loose, ops.NR_BF16), ops.pack(loose, ops.NR_BF16)
    assert p1.data_ptr() != p2.data_ptr()                                        # not in a bucket: packed afresh each time


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_flat_bucket_model_follows_load_state_dict(dt):
    """The Q|K|V operand of a bucketed model is a VIEW of the flat parameter buffer; `p.data = view` leaves every Parameter
    its own version counter, so `load_state_dict` (resume, or before eval) bumps neither the buffer's counter nor
    ops.param_epoch.  The packed copies must follow anyway (their stamp carries every bucket parameter's version): the
    forward after the load equals a fresh model built from those weights -- in eval mode, where no Adam step ever
    refreshes anything."""
    m, batches = _tiny_nrms(dt)
    fb = parallel.FlatBucket(m, lr=1e-3)
    m.eval()
    with torch.no_grad():
        before = m(*batches[0])[1].clone()
    torch.manual_seed(77)
    other, _ = _tiny_nrms(dt)
    with torch.no_grad():
        for p in other.parameters():
            if p.dim() > 0 and p.shape[0] != 200:                # every weight but the word table
                p.add_(torch.randn_like(p) * 0.05)
    sd = {k: v.detach().clone() for k, v in other.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    other.eval()
    with torch.no_grad():
        got, want = m(*batches[0])[1], other(*batches[0])[1]
    assert float((want - before).abs().max()) > 1e-3              # the new weights do change the scores
    assert torch.equal(got, want)
    assert all(p.data_ptr() == v[0].data_ptr() for p, v in zip(fb.params, fb.views))      # still views of the bucket


def test_zero_gradient_hint_is_dropped_when_autograd_adds_another_consumer():
    """ops._offer_flags / _take_flags: the pooling backward tells the NEXT libnrhip backward which sequences have an exactly
    zero upstream gradient.  With a second, torch-native consumer of the MHSA output autograd may add that gradient into
    the pooling's dx IN PLACE (same pointer, same size): the flags are stale then and must not be taken."""
    from newsrecommendation_amd.model.model_utils import AttentionPooling, MultiHeadSelfAttention
    torch.manual_seed(3)
    n, L, dm, h, d = 256, 30, 64, 20, 20
    mh = MultiHeadSelfAttention(dm, h, d, d, compute_dtype="bf16").cuda()
    pool = AttentionPooling(h * d, 200, compute_dtype="bf16").cuda()
    x = (torch.randn(n, L, dm, device="cuda") * 0.5).to(torch.bfloat16)
    gate = torch.zeros(n, 1, device="cuda")
    gate[::4] = 1.0                                               # 3 of 4 pooled vectors get a zero gradient
    grads = []
    for extra in (False, True):
        for p in list(mh.parameters()) + list(pool.parameters()):
            p.grad = None
        y = mh(x)
        out = (pool(y) * gate).sum()
        if extra:
            out = out + y.float().mean(1).sum() * 0.0 + (y.float().mean(1) * (1 - gate)).sum()
        out.backward()
        grads.append(mh.W_V.weight.grad.clone())
    # reference for the second case by linearity: d/dW of the extra term alone, from a run without the pooling
    for p in mh.parameters():
        p.grad = None
    y = mh(x)
    (y.float().mean(1) * (1 - gate)).sum().backward()
    only_extra = mh.W_V.weight.grad.clone()
    want = grads[0] + only_extra
    err = float((grads[1] - want).abs().max())
    assert float(only_extra.abs().max()) > 0
    assert err <= 2e-2 * float(want.abs().max()) + 1e-4, (err, float(want.abs().max()))


def test_stack_rows_and_split_rows_equal_the_torch_plumbing_they_replace():
    """`ops.stack_rows` = torch.cat of the candidate / history id rows + the `needed` flags Model.forward builds from the
    history mask; `ops.split_rows` = torch.split whose backward assembles the encoder-output gradient in place: with the
    pad-doc blend consuming the history half and the scorer the candidate half it returns the SAME buffer both wrote into
    (no concatenation), with any other consumer it concatenates like torch."""
    g = torch.Generator().manual_seed(3)
    B, C, H, T, N = 6, 5, 50, 30, 400
    cand = torch.randint(0, 99, (B, C, T), generator=g, dtype=torch.int32).cuda()
    hist = torch.randint(0, 99, (B, H, T), generator=g, dtype=torch.int32).cuda()
    mask = (torch.rand(B, H, generator=g) < 0.6).float().cuda()
    ids, flags = ops.stack_rows(cand, hist, mask)
    assert torch.equal(ids, torch.cat([cand.reshape(-1, T), hist.reshape(-1, T)]))
    assert flags.dtype == torch.int32 and torch.equal(flags, torch.cat([torch.ones(B * C, device="cuda"), mask.reshape(-1)]).int())
    assert ops.needed_flags(flags) is flags
    assert ops.stack_rows(cand, hist, None, flags=False)[1] is None

    vecs0 = (torch.randn(B * (C + H), N, generator=g) * 0.3).cuda()
    pad = (torch.randn(1, N, generator=g) * 0.3).cuda().requires_grad_(True)
    label = torch.randint(0, C, (B,), generator=g).cuda()

    def run(split):
        v = vecs0.clone().requires_grad_(True)
        pad.grad = None
        a, b = split(v)
        x = ops.pad_blend(b.reshape(B, H, N), mask, pad, ops.NR_F32)
        user = x.float().mean(dim=1)
        loss, score = ops.score_ce(a.reshape(B, C, N), user, label)
        loss.backward()
        return v.grad, pad.grad.clone(), float(loss)

    seen = {}
    orig = ops.SplitRowsFunction.backward

    def spy(ctx, ga, gb):
        out = orig(ctx, ga, gb)
        seen["same"] = ctx.arena is not None and out[0].data_ptr() == ctx.arena.data_ptr()
        return out
    ops.SplitRowsFunction.backward = staticmethod(spy)
    try:
        got = run(lambda v: ops.split_rows(v, B * C))
    finally:
        ops.SplitRowsFunction.backward = staticmethod(orig)
    want = run(lambda v: v.split([B * C, B * H], dim=0))
    assert seen["same"]                                   # both halves were written in place: nothing was concatenated
    assert torch.equal(got[0], want[0]) and got[2] == want[2]
    assert torch.allclose(got[1], want[1], rtol=1e-5, atol=1e-7)          # (d pad_doc: fp32 atomics, the order varies)

    # another consumer of a half: the gradient arrives in some other tensor and is concatenated
    v = vecs0.clone().requires_grad_(True)
    a, b = ops.split_rows(v, B * C)
    (a.sum() * 2 + (b * b).sum()).backward()
    assert torch.equal(v.grad[:B * C], torch.full_like(v.grad[:B * C], 2.0)) and torch.allclose(v.grad[B * C:], 2 * vecs0[B * C:])


def test_adam_kernel_repacks_the_word_table_in_the_same_pass():
    """A bucketed embedding table that has a current packed bf16 copy (ops.table_cache) gets the new values written into
    that copy by nr_adam_step_packed itself: after the step the cache still serves the SAME buffer, it holds exactly
    bf16(new fp32 values) with zero padding columns, and no nr_cast_pad of the table is launched.  A torch-side write
    afterwards still invalidates it (version counter)."""
    from newsrecommendation_amd import _lib
    torch.manual_seed(1)
    emb = torch.nn.Embedding(3001, 300, padding_idx=0).cuda()                    # 900 300 elements: above the small-weight cache
    lin = torch.nn.Linear(300, 64).cuda()
    net = torch.nn.Sequential(emb, lin)
    fb = parallel.FlatBucket(net, lr=1e-2)
    w = emb.weight
    pk0 = ops.table_cache.get(w, ops.NR_BF16)
    assert pk0.shape == (3001, 320) and torch.equal(pk0[:, :300], w.detach().to(torch.bfloat16))
    for step in range(2):
        fb.grad.copy_(torch.randn(fb.numel, device="cuda") * 0.1)
        _lib.prof_enable(1)
        try:
            _lib.prof_collect()
            fb.adam_step()
            pk = ops.table_cache.get(w, ops.NR_BF16)
            torch.cuda.synchronize()
            labels = set(_lib.prof_collect().keys())
        finally:
            _lib.prof_enable(0)
        assert pk.data_ptr() == pk0.data_ptr()
        assert not any(l.startswith("cast_pad[") for l in labels), labels
        assert torch.equal(pk[:, :300], w.detach().to(torch.bfloat16)) and float(pk[:, 300:].abs().sum()) == 0.0
    ref = torch.optim.Adam([torch.zeros(1)], lr=1e-2)                             # (the update itself is covered by the tests above)
    with torch.no_grad():
        w.mul_(0.5)
    pk2 = ops.table_cache.get(w, ops.NR_BF16)
    assert torch.equal(pk2[:, :300], w.detach().to(torch.bfloat16))
