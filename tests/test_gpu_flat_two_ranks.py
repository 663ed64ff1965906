"""GPU, world_size 2: the DEFAULT data-parallel mode (`dp_mode="flat"`) of `train.train()` with more than one rank
(src/main.py:82,108-110 semantics: rank-0 parameter broadcast, gradient mean over ranks, Adam).

Two FRESH child processes (never a re-exec of a process that touched the GPU), both on cuda:0, `gloo` process group
(one-GPU boxes: RCCL refuses two ranks on one device; the collective SEQUENCE is the same), real `NRMS.Model` in bf16 with
a trainable 5 000 x 300 word table (>= 2**20 elements, so the bucket lays it out last and arms the early all-reduce),
dropout on, B = 16 per rank, 3 steps.  Checked, step by step (an end-to-end trajectory comparison is ill-posed: Adam turns
the last-bit noise of the fp32 atomics on gradient elements of ~1e-7 into parameter differences of ~1e-6 after ONE step, bf16
rounding of the weights then amplifies them -- measured 1e-6 / 2e-4 / 1e-3 after steps 1 / 2 / 3, lr 1e-3):

  * both ranks end with bit-identical flat parameter buffers (they started from different seeds: rank 0's must have won),
    and every step starts from bit-identical parameters on both ranks;
  * the early hook fired exactly once per backward (`FlatBucket.early_calls`), the table sits last in the bucket;
  * per step, a single-process replay STARTING FROM THAT STEP'S PARAMETERS -- each rank's batch with that rank's own
    dropout-seed stream -- reproduces each rank's local gradient (<= 1e-5 * max|g|: the order of the fp32 atomics; 2.8e-6 seen) and loss;
  * the gradient both ranks hold after the split collective (early table all-reduce + the rest) is EXACTLY the fp32 sum of
    the two local gradients, table part and front part alike;
  * `nr_adam_step` with grad_scale 1 on the mean of that sum gives the next step's parameters BIT FOR BIT (what the
    ranks computed with grad_scale 1/2 on the sum: the power-of-two factor commutes with the rounding);
  * a second backward before `step()` raises, and `zero_grad()` clears the pending early all-reduce."""
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
T, H, K, V, B, STEPS = 30, 50, 4, 5000, 16, 3


def _args(tmp):
    return SimpleNamespace(model="NRMS", num_words_title=T, user_log_length=H, npratio=K, word_embedding_dim=300, news_dim=400,
                           num_attention_heads=20, news_query_vector_dim=200, user_query_vector_dim=200, drop_rate=0.2,
                           user_log_mask=False, freeze_embedding=False, use_category=False, use_subcategory=False,
                           category_emb_dim=100, compute_dtype="bf16", lr=1e-3, batch_size=B, epochs=1, log_steps=1000, nGPU=2,
                           dp_mode="flat", feed="device", model_dir=None, train_data_dir=os.path.join(tmp, "train"))


def _world(tmp, n_news=400, n_imp=150, seed=7):
    rnd = random.Random(seed)
    news_ids = [f"N{i}" for i in range(1, n_news + 1)]
    news_index = {nid: i + 1 for i, nid in enumerate(news_ids)}
    g = torch.Generator().manual_seed(seed)
    comb = torch.randint(1, V, (n_news + 1, T), generator=g, dtype=torch.int32)
    comb[0] = 0
    for r in range(1, n_news + 1):
        comb[r, rnd.randint(5, T):] = 0
    table = torch.randn(V, 300, generator=g) * 0.4
    table[0] = 0
    lines = []
    for i in range(n_imp):
        hist = " ".join(rnd.choice(news_ids) for _ in range(rnd.randint(0, 60)))
        imps = [f"{rnd.choice(news_ids)}-{1 if j == 0 else 0}" for j in range(rnd.randint(3, 9))]
        lines.append("\t".join([str(i + 1), "U1", "t", hist, " ".join(imps)]) + "\n")
    os.makedirs(os.path.join(tmp, "train"), exist_ok=True)
    with open(os.path.join(tmp, "train", "behaviors.tsv"), "w") as f:
        f.writelines(lines)
    np.savez(os.path.join(tmp, "world.npz"), comb=comb.numpy(), table=table.numpy())
    with open(os.path.join(tmp, "news_index.json"), "w") as f:
        json.dump(news_index, f)
    return news_index, comb.numpy(), table.numpy()


_CHILD = r"""
import json, os, random, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["NR_ROOT"], "tests"))
from test_gpu_flat_two_ranks import _args, STEPS
from newsrecommendation_amd import parallel as P, train as TR
rank, tmp = int(os.environ["RANK"]), os.environ["NR_TMP"]
dist.init_process_group("gloo", init_method="env://", world_size=2, rank=rank)
made = []
_init = P.FlatBucket.__init__
def init(self, *a, **k):
    _init(self, *a, **k)
    made.append(self)
P.FlatBucket.__init__ = init
_early = P.FlatBucket._early_allreduce
def early(self):                        # the table gradient as THIS rank computed it: once the hook has fired, the in-place
    self._local_tail = self.grad[self._big_off:].clone()      # all-reduce may overwrite it at any moment
    _early(self)
P.FlatBucket._early_allreduce = early
def step(self):                         # FlatBucket.step with the tensors of every stage written out
    i = self.t
    local = self.grad.clone()           # (the front part is reduced only inside allreduce() below)
    local[self._big_off:] = self._local_tail
    torch.save({"param": self.param.cpu(), "local": local.cpu()}, os.path.join(tmp, f"rank{rank}_step{i}.pt"))
    self.allreduce()
    torch.save(self.grad.cpu(), os.path.join(tmp, f"rank{rank}_step{i}_reduced.pt"))
    self.adam_step(zero_grad=True)
P.FlatBucket.step = step
z = np.load(os.path.join(tmp, "world.npz"))
news_index = json.load(open(os.path.join(tmp, "news_index.json")))
args = _args(tmp)
torch.manual_seed(100 + rank)          # different initial weights per rank: the bucket must broadcast rank 0's
random.seed(50 + rank)                 # label positions (SURVEY App. C.9: the reference leaves them unseeded per rank)
model, losses = TR.train(rank, args, news_index, z["comb"], z["table"], device="cuda:0", max_steps=STEPS, log=lambda *_: None)
fb = made[0]
out = {"losses": losses.tolist(), "early_calls": fb.early_calls, "big_off": fb._big_off, "numel": fb.numel, "t": fb.t,
       "table_last": fb.params[-1] is model.news_encoder.embedding_matrix.weight}
torch.save({"param": fb.param.cpu(), "sd": {k: v.detach().cpu() for k, v in model.state_dict().items()}},
           os.path.join(tmp, f"rank{rank}.pt"))
# one backward per optimizer step: a second one before step() must raise (the table gradient is already on the wire)
from newsrecommendation_amd.data import IndexedTrainShard
feed = TR.DeviceFeed(IndexedTrainShard(os.path.join(args.train_data_dir, f"behaviors_np{args.npratio}_{rank}.tsv"), news_index, args),
                     z["comb"], args.batch_size, torch.device("cuda:0"))
feed.start_epoch()
batch = feed.batch(0)
model(*batch)[0].backward()
try:
    model(*batch)[0].backward()
    out["second"] = ""
except RuntimeError as e:
    out["second"] = str(e)
fb.zero_grad()
out["cleared"] = fb._early_work is None
out["early_calls_after"] = fb.early_calls      # the accepted extra backward counts, the rejected one does not
torch.cuda.synchronize()
dist.barrier()
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_flat_bucket_two_ranks_equal_a_single_process_replay(tmp_path):
    from newsrecommendation_amd import parallel as P, train as TR
    tmp = str(tmp_path)
    news_index, comb, table = _world(tmp)
    args = _args(tmp)
    n = D.prepare_training_data(args.train_data_dir, 2, K, seed=1)
    assert n >= 2 * B * STEPS
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, PYTHONPATH=ROOT, NR_ROOT=ROOT, NR_TMP=tmp, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        logs = [open(os.path.join(tmp, f"rank{r}.{s}"), "w") for s in ("out", "err")]       # files, not pipes: nobody blocks
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD], env=env, cwd=ROOT, stdout=logs[0], stderr=logs[1]))
    outs = []
    for r, p in enumerate(procs):
        try:
            p.wait(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        so, se = (open(os.path.join(tmp, f"rank{r}.{s}")).read() for s in ("out", "err"))
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    r0, r1 = (torch.load(os.path.join(tmp, f"rank{r}.pt"), weights_only=True) for r in range(2))

    # the split collective was armed and ran once per backward on both ranks; both hold the same parameters, bit for bit
    for o in outs:
        assert o["table_last"] and o["big_off"] is not None and o["numel"] - o["big_off"] >= V * 300
        assert o["early_calls"] == STEPS and o["early_calls_after"] == STEPS + 1 and o["t"] == STEPS and len(o["losses"]) == STEPS
        assert "second backward" in o["second"] and o["cleared"]
    assert torch.equal(r0["param"], r1["param"])
    assert all(torch.equal(r0["sd"][k], r1["sd"][k]) for k in r0["sd"])

    # single-process replay, step by step from the ranks' own parameters
    dev = torch.device("cuda:0")
    rng, feeds, model = [], [], None
    for r in range(2):
        torch.manual_seed(100 + r)
        m = TR.build_model(args, table)                      # consumes the same init draws as the child did
        rng.append(torch.get_rng_state())
        if r == 0:
            model = m.to(dev)
        random.seed(50 + r)
        feed = TR.DeviceFeed(D.IndexedTrainShard(os.path.join(args.train_data_dir, f"behaviors_np{K}_{r}.tsv"), news_index, args),
                             comb, B, dev)
        feed.start_epoch()
        feeds.append(feed)
    fb = P.FlatBucket(model, lr=args.lr)
    model.train()
    load = lambda name: torch.load(os.path.join(tmp, name), weights_only=True)
    for i in range(STEPS):
        rec = [load(f"rank{r}_step{i}.pt") for r in range(2)]
        assert torch.equal(rec[0]["param"], rec[1]["param"]), i                   # every step starts from the same parameters
        if i == 0:
            assert torch.equal(rec[0]["param"], fb.param.cpu())                  # ... the first from rank 0's initial weights
        fb.param.copy_(rec[0]["param"].to(dev))
        fb._params_changed()
        for r in range(2):
            torch.set_rng_state(rng[r])
            loss, _ = model(*feeds[r].batch(i))
            loss.backward()
            rng[r] = torch.get_rng_state()
            assert abs(float(loss.detach()) - outs[r]["losses"][i]) <= 2e-6 * max(1.0, abs(float(loss.detach()))), (i, r)
            g, want = fb.grad.cpu(), rec[r]["local"]
            assert float(want.abs().max()) > 1e-3
            assert float((g - want).abs().max()) <= 1e-5 * float(want.abs().max()), (i, r, float((g - want).abs().max()))
            fb.zero_grad()
        reduced = [load(f"rank{r}_step{i}_reduced.pt") for r in range(2)]
        assert torch.equal(reduced[0], reduced[1]) and torch.equal(reduced[0], rec[0]["local"] + rec[1]["local"]), i
        fb.grad.copy_((reduced[0] * 0.5).to(dev))            # the mean over ranks; this bucket's Adam kernel runs with grad_scale 1
        fb.adam_step()
        torch.cuda.synchronize()
        nxt = load(f"rank0_step{i + 1}.pt")["param"] if i + 1 < STEPS else r0["param"]
        assert torch.equal(fb.param.cpu(), nxt), (i, float((fb.param.cpu() - nxt).abs().max()))
        assert float((nxt - rec[0]["param"]).abs().max()) >= 0.5 * args.lr      # (and the step did move the parameters)
