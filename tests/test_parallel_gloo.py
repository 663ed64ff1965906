"""CPU, world_size 2, gloo: the multi-process path of the PRODUCT loops.

`train.train()` / `train.test()` themselves run here as two ranks -- process-group creation (env://), shard file
selection by rank, the agreed batch count (unequal shards must not hang), the DistributedDataParallel wrap with its
rank-0 broadcast and gradient mean, the sharded corpus encode + all_gather, and the SUM-reduce of the metric sums with
the reference's divisor -- with a small CPU nn.Module standing in for the HIP-backed model (the HIP ops need a GPU;
their N=1 numerics are covered by the -m gpu tests).  `parallel.FlatBucket`'s collective half (broadcast, one
all-reduce over the flat gradient buffer) runs on CPU tensors too; its Adam kernel is GPU-only and must say so."""
import os
import random
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from newsrecommendation_amd import data as D, metrics as M, parallel as P, train as TR

T, H, K, DIM = 6, 4, 2, 8


class StandIn(torch.nn.Module):
    """Same call surface as model.NRMS.Model: forward(history, mask, candidate, label) -> (loss, score), news_encoder,
    user_encoder, args.  Plain torch ops."""

    def __init__(self, args, embedding_matrix, *unused):
        super().__init__()
        self.args = args
        self.emb = torch.nn.Embedding.from_pretrained(torch.from_numpy(embedding_matrix).float(), freeze=False, padding_idx=0)
        self.proj = torch.nn.Linear(embedding_matrix.shape[1], DIM)
        self.user = torch.nn.Linear(DIM, DIM)

    def news_encoder(self, ids):
        return torch.tanh(self.proj(self.emb(ids.long()).mean(-2)))

    def user_encoder(self, vecs, mask):
        m = mask.unsqueeze(-1)
        return self.user((vecs * m).sum(1) / (m.sum(1) + 1.0))

    def forward(self, history, history_mask, candidate, label):
        cv, hv = self.news_encoder(candidate), self.news_encoder(history)
        score = torch.bmm(cv, self.user_encoder(hv, history_mask).unsqueeze(-1)).squeeze(-1)
        return torch.nn.functional.cross_entropy(score, label), score


def _world(tmp, n_lines):
    rnd = random.Random(3)
    ids = [f"N{i}" for i in range(1, 31)]
    news_index = {n: i + 1 for i, n in enumerate(ids)}
    g = torch.Generator().manual_seed(3)
    comb = torch.randint(1, 40, (31, T), generator=g, dtype=torch.int32).numpy()
    comb[0] = 0
    table = (torch.randn(40, 5, generator=g) * 0.5).numpy()
    lines = []
    for i in range(n_lines):
        hist = " ".join(rnd.choice(ids) for _ in range(rnd.randint(0, 6)))
        imps = " ".join(f"{rnd.choice(ids)}-{1 if j == 0 or rnd.random() < 0.2 else 0}" for j in range(rnd.randint(3, 7)))
        lines.append("\t".join([str(i), "U", "t", hist, imps]) + "\n")
    for sub in ("train", "test"):
        os.makedirs(os.path.join(tmp, sub), exist_ok=True)
        with open(os.path.join(tmp, sub, "behaviors.tsv"), "w") as f:
            f.writelines(lines)
    args = SimpleNamespace(model="StandIn", user_log_length=H, npratio=K, num_words_title=T, news_dim=DIM, lr=1e-2, batch_size=8, epochs=1,
                           log_steps=1000, nGPU=2, dp_mode="ddp", feed="host", model_dir=None, shard_encode=True,
                           train_data_dir=os.path.join(tmp, "train"), test_data_dir=os.path.join(tmp, "test"))
    return args, news_index, comb, table


def _numpy_scorer(model, news_vecs, shard, batch_size, device):
    """score_fn stand-in for the device scorer: same contract (scores in CSR order, [scored, 4 sums])."""
    hist, mask = torch.from_numpy(shard.hist).long(), torch.from_numpy(shard.mask)
    user = model.user_encoder(news_vecs[hist], mask)
    sums, scores = np.zeros(5), []
    for i in range(len(shard)):
        a, b = shard.offsets[i], shard.offsets[i + 1]
        s = (news_vecs[torch.from_numpy(shard.cand[a:b]).long()] @ user[i]).numpy()
        scores.append(s)
        y = shard.label[a:b]
        if y.mean() in (0, 1):
            continue
        sums += [1, M.roc_auc_score(y, s), M.mrr_score(y, s), M.ndcg_score(y, s, 5), M.ndcg_score(y, s, 10)]
    return torch.from_numpy(np.concatenate(scores)), sums


def _rank_main(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    args, news_index, comb, table = _world(tmp, 0)            # files already written by the parent
    torch.manual_seed(100 + rank)                             # different initial weights per rank: DDP must broadcast rank 0's
    random.seed(50 + rank)
    logs = []
    model, losses = TR.train(rank, args, news_index, comb, table, device="cpu", model_factory=StandIn, log=logs.append)
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    n_samples, means = TR.test(rank, args, model, news_index, comb, device="cpu", score_fn=_numpy_scorer, log=logs.append)
    # the flat bucket's collective half on the same process group
    torch.manual_seed(7 + rank)
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    fb = P.FlatBucket(lin, lr=1e-2)
    w0 = fb.param.clone()
    lin(torch.full((2, 4), float(rank + 1))).sum().backward()
    g_local = fb.grad.clone()
    fb.allreduce()
    try:
        fb.adam_step()
        adam_err = ""
    except RuntimeError as e:
        adam_err = str(e)
    # the SPLIT collective of the default mode at world > 1: a parameter of >= 2**20 elements is laid out last, its gradient
    # is all-reduced early (async, from the backward's hook), everything in front of it by allreduce()
    torch.manual_seed(9 + rank)
    big = torch.nn.Sequential(torch.nn.Embedding(4096, 256), torch.nn.Linear(256, 3))
    fb2 = P.FlatBucket(big, lr=1e-2)
    table = big[0].weight
    split = {"armed": hasattr(table, "_nr_grad_ready"), "big_off": fb2._big_off, "numel": fb2.numel}
    ids = torch.arange(64).reshape(8, 8) * (rank + 3) % 4096
    big(ids).mul(float(rank + 1)).sum().backward()
    g2_local = fb2.grad.clone()
    table._nr_grad_ready()                                    # what ops.MHSAFunction.backward does after the table-gradient kernel
    try:
        table._nr_grad_ready()
        split["second"] = ""
    except RuntimeError as e:
        split["second"] = str(e)
    fb2.allreduce()
    g2_split = fb2.grad.clone()
    fb2.grad.copy_(g2_local)
    fb2.allreduce()                                           # no hook call this time: ONE all-reduce over the whole buffer
    split["same"] = bool(torch.equal(g2_split, fb2.grad))
    table._nr_grad_ready()                                    # an aborted step: zero_grad() must wait for and drop the early handle
    fb2.zero_grad()
    split["cleared"] = fb2._early_work is None and float(fb2.grad.abs().sum()) == 0.0
    split["calls"] = fb2.early_calls                          # (the rejected second announcement is not counted)
    nz = torch.nonzero(g2_local).flatten()
    q.put((rank, len(losses), w.tolist(), n_samples, means.tolist(), w0.tolist(), g_local.tolist(), fb.grad.tolist(), adam_err,
           split, (nz.tolist(), g2_local[nz].tolist()), (torch.nonzero(g2_split).flatten().tolist(), g2_split[torch.nonzero(g2_split).flatten()].tolist())))
    dist.barrier()
    dist.destroy_process_group()


def test_train_and_test_loops_as_two_gloo_ranks(tmp_path):
    tmp = str(tmp_path)
    args, news_index, comb, table = _world(tmp, 140)
    n = D.prepare_training_data(args.train_data_dir, 2, K, seed=1)
    D.prepare_testing_data(args.test_data_dir, 2)
    sizes = [sum(1 for _ in open(os.path.join(args.train_data_dir, f"behaviors_np{K}_{r}.tsv"))) for r in range(2)]
    assert sum(sizes) == n and sizes[0] - sizes[1] in (0, 1)
    world, port = 2, 29500 + os.getpid() % 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, w_a, ns_a, means_a, w0_a, gl_a, g_a, err_a, sp_a, l2_a, s2_a), (_, n1, w_b, ns_b, means_b, w0_b, gl_b, g_b, err_b, sp_b, l2_b, s2_b) = res
    # both ranks ran the agreed number of batches and hold identical parameters
    expect = min((s + args.batch_size - 1) // args.batch_size for s in sizes)
    assert n0 == n1 == expect
    assert w_a == w_b

    # single-process replay of the same job: rank-0 init, per-step gradient = mean of the two ranks' batch gradients
    torch.manual_seed(100)
    ref = StandIn(args, table)
    opt = torch.optim.Adam(ref.parameters(), lr=args.lr)
    streams = []
    for r in range(2):
        random.seed(50 + r)
        ds = D.DatasetTrain(os.path.join(args.train_data_dir, f"behaviors_np{K}_{r}.tsv"), news_index, comb, args)
        streams.append(list(torch.utils.data.DataLoader(ds, batch_size=args.batch_size))[:expect])
    for b0, b1 in zip(*streams):
        opt.zero_grad()
        (0.5 * (ref(*b0)[0] + ref(*b1)[0])).backward()
        opt.step()
    w_ref = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    assert torch.allclose(torch.tensor(w_a), w_ref, atol=2e-6), float((torch.tensor(w_a) - w_ref).abs().max())

    # eval: sharded encode + all_gather, per-rank scoring, one SUM reduce; distributed divisor = ALL samples (src/main.py:269-273)
    assert ns_a == 140
    ref.eval()
    with torch.no_grad():
        torch.nn.utils.vector_to_parameters(torch.tensor(w_a), ref.parameters())     # rank metrics are discontinuous: same weights
        nv = ref.news_encoder(torch.from_numpy(comb))
        sums = np.zeros(5)
        for r in range(2):
            shard = D.IndexedTestShard(os.path.join(args.test_data_dir, f"behaviors_{r}.tsv"), news_index, args)
            sums += _numpy_scorer(ref, nv, shard, 8, "cpu")[1]
    assert 0 < sums[0] <= 140
    assert np.allclose(means_a, sums[1:] / 140, atol=1e-6), (means_a, sums[1:] / 140)

    # FlatBucket: rank-0 broadcast at construction, ONE all-reduce (SUM) over the flat gradient buffer, GPU-only Adam
    assert w0_a == w0_b
    assert np.allclose(np.array(g_a), np.array(gl_a) + np.array(gl_b)) and g_a == g_b
    assert "no CPU fallback" in err_a and "no CPU fallback" in err_b

    # the split collective (early table all-reduce + the rest): armed on both ranks, the table laid out last, result = sum of
    # the local gradients = what the single all-reduce gives; a second announcement before step() raises; zero_grad() clears
    for sp in (sp_a, sp_b):
        assert sp["armed"] and 0 < sp["big_off"] and sp["numel"] - sp["big_off"] >= 4096 * 256
        assert "second backward" in sp["second"] and sp["same"] and sp["calls"] == 2 and sp["cleared"]
    want = {}
    for idx, val in (l2_a, l2_b):
        for i, v in zip(idx, val):
            want[i] = want.get(i, 0.0) + v
    for idx, val in (s2_a, s2_b):
        got = dict(zip(idx, val))
        assert set(got) == {i for i, v in want.items() if v != 0.0}
        assert all(abs(got[i] - want[i]) <= 1e-6 * max(1.0, abs(want[i])) for i in got)


def test_agreed_batch_count_single_process_and_sharding():
    assert P.agree_on_batches(7, "cpu") == 7
    lines = [f"{i}\tU1\tt\tN1 N2\tN3-1 N4-0 N5-0\n" for i in range(101)]
    shards = D.shard_training_lines(lines, 2, 1, 0)
    assert [len(s) for s in shards] == [51, 50] and set(shards[0]).isdisjoint(shards[1])
    assert D.common_batch_count([len(s) for s in shards], 8) == 6
    t = D.shard_testing_lines(lines, 2)
    assert t[0] == lines[0::2] and t[1] == lines[1::2]
