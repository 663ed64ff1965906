"""GPU, world_size 2: the evaluation job with more than one rank (src/main.py:145-277 semantics): every rank
encodes a contiguous half of the news corpus, the halves are all-gathered into the full news-vector table, every rank scores
its own behaviour shard (`behaviors_{rank}.tsv`) and the metric sums are reduced (`parallel.reduce_eval_sums`).

Two FRESH child processes on cuda:0 over `gloo` (one-GPU boxes: RCCL refuses two ranks on one device; the collective sequence is
the same), real `NRMS.Model` in bf16 with the masked user encoder (src/demo.sh:26), 400 news, 160 impressions with histories of
0..60 clicks (front-padded / truncated to 50: both the 32-slot and the 50-slot user paths run).  Checked against ONE process
that encodes the whole corpus itself and scores both shards:

  * the scores of every impression of both ranks (bit-identical news vectors -> <= 1e-6 abs),
  * on rank 0 (the reduction's destination): n_samples = all impressions, means = reduced sums / reduced scored count
    (`eval_divide_by_scored=True`) <= 1e-9."""
import json
import os
import random
import socket
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from newsrecommendation_amd import data as D

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, H, V, N_NEWS, N_IMP = 30, 50, 3000, 400, 160


def _args(tmp):
    return SimpleNamespace(model="NRMS", num_words_title=T, user_log_length=H, npratio=4, word_embedding_dim=300, news_dim=400,
                           num_attention_heads=20, news_query_vector_dim=200, user_query_vector_dim=200, drop_rate=0.2,
                           user_log_mask=True, freeze_embedding=False, use_category=False, use_subcategory=False,
                           category_emb_dim=100, compute_dtype="bf16", batch_size=64, nGPU=2, shard_encode=True,
                           eval_divide_by_scored=True, test_data_dir=os.path.join(tmp, "test"))


def _world(tmp, seed=11):
    rnd = random.Random(seed)
    news_ids = [f"N{i}" for i in range(1, N_NEWS + 1)]
    news_index = {nid: i + 1 for i, nid in enumerate(news_ids)}
    g = torch.Generator().manual_seed(seed)
    comb = torch.randint(1, V, (N_NEWS + 1, T), generator=g, dtype=torch.int32)
    comb[0] = 0
    for r in range(1, N_NEWS + 1):
        comb[r, rnd.randint(5, T):] = 0
    table = torch.randn(V, 300, generator=g) * 0.4
    table[0] = 0
    lines = []
    for i in range(N_IMP):
        hist = " ".join(rnd.choice(news_ids) for _ in range(rnd.randint(0, 60)))
        imps = [f"{rnd.choice(news_ids)}-{1 if (j == 0 or rnd.random() < 0.15) else 0}" for j in range(rnd.randint(2, 12))]
        lines.append("\t".join([str(i + 1), "U1", "t", hist, " ".join(imps)]) + "\n")
    os.makedirs(os.path.join(tmp, "test"), exist_ok=True)
    with open(os.path.join(tmp, "test", "behaviors.tsv"), "w") as f:
        f.writelines(lines)
    np.savez(os.path.join(tmp, "world.npz"), comb=comb.numpy(), table=table.numpy())
    with open(os.path.join(tmp, "news_index.json"), "w") as f:
        json.dump(news_index, f)
    return news_index, comb.numpy(), table.numpy()


_CHILD = r"""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["NR_ROOT"], "tests"))
from test_gpu_eval_two_ranks import _args
from newsrecommendation_amd import train as TR
rank, tmp = int(os.environ["RANK"]), os.environ["NR_TMP"]
dist.init_process_group("gloo", init_method="env://", world_size=2, rank=rank)
z = np.load(os.path.join(tmp, "world.npz"))
news_index = json.load(open(os.path.join(tmp, "news_index.json")))
args = _args(tmp)
torch.manual_seed(0)                                   # the same weights on both ranks (a checkpoint, in a real job)
model = TR.build_model(args, z["table"]).to("cuda:0")
got = []
n_samples, means = TR.test(rank, args, model, news_index, z["comb"], log=lambda *_: None, collect_scores=got, device="cuda:0")
np.savez(os.path.join(tmp, f"rank{rank}.npz"), scores=np.concatenate([s for _, s in got]), labels=np.concatenate([l for l, _ in got]),
         counts=np.asarray([len(s) for _, s in got]), means=np.asarray(means), n_samples=n_samples)
torch.cuda.synchronize()
dist.barrier()
print("RESULT ok")
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_encode_and_reduced_metrics_equal_a_single_process(tmp_path):
    from newsrecommendation_amd import train as TR
    tmp = str(tmp_path)
    news_index, comb, table = _world(tmp)
    args = _args(tmp)
    D.prepare_testing_data(args.test_data_dir, 2)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, PYTHONPATH=ROOT, NR_ROOT=ROOT, NR_TMP=tmp, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        logs = [open(os.path.join(tmp, f"rank{r}.{s}"), "w") for s in ("out", "err")]
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD], env=env, cwd=ROOT, stdout=logs[0], stderr=logs[1]))
    for r, p in enumerate(procs):
        try:
            p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, open(os.path.join(tmp, f"rank{r}.err")).read()[-3000:]

    # one process: the whole corpus encoded here, both shards scored here
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = TR.build_model(args, table).to(dev).eval()
    nv = TR.encode_news(model, comb, args.batch_size, dev)
    sums_all, n_all = np.zeros(5), 0
    for r in range(2):
        z = np.load(os.path.join(tmp, f"rank{r}.npz"))
        shard = D.IndexedTestShard(os.path.join(args.test_data_dir, f"behaviors_{r}.tsv"), news_index, args)
        assert len(shard) > 40 and (shard.mask.sum(1) <= 32).any() and (shard.mask.sum(1) > 32).any()   # both user paths
        scores, sums = TR.score_shard(model, nv, shard, args.batch_size, dev)
        assert np.array_equal(z["labels"], shard.label) and np.array_equal(z["counts"], shard.offsets[1:] - shard.offsets[:-1])
        assert np.allclose(z["scores"], scores.cpu().numpy(), atol=1e-6), float(np.abs(z["scores"] - scores.cpu().numpy()).max())
        sums_all += np.asarray(sums.cpu().tolist())
        n_all += len(shard)
    assert n_all == N_IMP and sums_all[0] > 60
    z = np.load(os.path.join(tmp, "rank0.npz"))           # the reduction's destination (src/main.py:269-273: dst = 0)
    assert int(z["n_samples"]) == N_IMP
    assert np.allclose(z["means"], sums_all[1:] / sums_all[0], atol=1e-9), (z["means"], sums_all[1:] / sums_all[0])
