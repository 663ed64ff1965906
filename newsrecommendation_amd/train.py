"""Train / eval loops with the shape of the reference's src/main.py (train :22-142, test :145-277).

File parsing of news.tsv and the title-embedding generation are out of scope: the loops take
the arrays those steps produce (`news_index`, `news_combined`, `embedding_matrix`).  Everything from the sharded
behaviours files onward follows the reference -- model by name, checkpoint dict layout, forward, acc, backward,
gradient all-reduce-mean, Adam, logging -- with the callers either side of the encoder path moved to the device
(SURVEY §8 rows f1, f2, f4):

  input   `args.feed = "device"` (default on the GPU): the shard is parsed ONCE into news-index arrays
          (`data.IndexedTrainShard`), uploaded, and every batch is assembled on the device (`ops.assemble_batch`);
          `"host"`: the reference's `DatasetTrain` + `DataLoader` (src/main.py:89-90), kept for comparison and for CPU runs.
  update  `args.dp_mode = "flat"` (default on the GPU): `parallel.FlatBucket` -- one gradient all-reduce, one fused
          HIP Adam kernel; `"ddp"`: `DistributedDataParallel` + `torch.optim.Adam` as src/main.py:76,82.
  eval    the [N+1, news_dim] news-vector table stays on the device (encode optionally sharded over the ranks +
          all_gather), history vectors are gathered there, scores and the per-impression AUC / MRR / nDCG are computed
          there (`ops.score_eval`, `ops.eval_metrics`); five numbers per rank leave the device and are SUM-reduced to
          rank 0 (src/main.py:269-273).

Every rank runs the same number of batches (`parallel.agree_on_batches`): the reference's shards differ by one sample
and its ranks can hang in the last all-reduce.
"""
import importlib
import itertools
import logging
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from . import ops, parallel
from .data import DatasetTrain, IndexedTestShard, IndexedTrainShard


def acc(y_true, y_hat):
    """src/utils.py:36-40."""
    y_hat = torch.argmax(y_hat, dim=-1)
    return (y_true == y_hat).sum().float() / y_true.shape[0]


def build_model(args, embedding_matrix, n_category=0, n_subcategory=0):
    """src/main.py:63-64: the model module is chosen by name."""
    module = importlib.import_module(f"newsrecommendation_amd.model.{args.model}")
    return module.Model(args, embedding_matrix, n_category, n_subcategory)


def checkpoint_dict(model, category_dict=None, subcategory_dict=None):
    """src/main.py:118-142 layout; DDP's 'module.' prefix stripped.  Tensors are copied to the HOST: a device-side clone
    would double a frozen multi-GB table in HBM at every epoch end, and a host copy also detaches the FlatBucket views from
    their shared buffer (each entry is saved with its own storage)."""
    def own(v):
        t = v.detach().cpu()
        return t.clone() if t.untyped_storage().nbytes() > t.numel() * t.element_size() else t      # (a view of a CPU bucket)
    sd = {(k[len("module."):] if k.startswith("module.") else k): own(v) for k, v in model.state_dict().items()}
    return {"model_state_dict": sd, "category_dict": category_dict or {}, "subcategory_dict": subcategory_dict or {}}


def load_checkpoint(path):
    """A checkpoint written by the reference (src/main.py:118-142) or by `train()`: tensors + plain dicts only, so the
    loader that executes nothing from the file suffices."""
    return torch.load(path, map_location="cpu", weights_only=True)


def _resolve_device(rank, is_distributed, device):
    if device is not None:
        return torch.device(device)
    return torch.device("cuda", rank if is_distributed else torch.cuda.current_device())


def _count_lines(path):
    with open(path, "rb") as f:
        return sum(1 for _ in f)


class DeviceFeed:
    """Row f1: the shard's index arrays and `news_combined` live on the device; `batch(i)` assembles batch i there."""

    def __init__(self, shard: IndexedTrainShard, news_combined, batch_size, device):
        comb = np.ascontiguousarray(news_combined).reshape(len(news_combined), -1)
        self.comb = torch.as_tensor(comb.astype(np.int32, copy=False), device=device)
        self.hist = torch.as_tensor(shard.hist, device=device)
        self.mask = torch.as_tensor(shard.mask, device=device)
        self.pos = torch.as_tensor(shard.pos, device=device)
        self.neg = torch.as_tensor(shard.neg, device=device)
        self.shard, self.B, self.n, self.device = shard, int(batch_size), len(shard), device
        self.label = None
        self.feat_shape = tuple(np.asarray(news_combined).shape[1:])

    def __len__(self):
        return (self.n + self.B - 1) // self.B                 # DataLoader default: the last partial batch is kept

    def start_epoch(self):
        self.label = torch.as_tensor(self.shard.draw_labels(), device=self.device)      # the epoch's random.randint stream

    def batch(self, i):
        a, b = i * self.B, min(self.n, (i + 1) * self.B)
        label = self.label[a:b]
        history, candidate = ops.assemble_batch(self.comb, self.hist[a:b], self.pos[a:b], self.neg[a:b], label)
        H, C = history.shape[1], candidate.shape[1]
        return (history.view(b - a, H, *self.feat_shape), self.mask[a:b], candidate.view(b - a, C, *self.feat_shape), label)


def train(rank, args, news_index, news_combined, embedding_matrix, category_dict=None, subcategory_dict=None,
          max_steps=None, log=logging.info, device=None, model_factory=None):
    """One rank of the training job (src/main.py:22-142).  rank=None: single process; otherwise rank `rank` of
    `args.nGPU` processes (the process group is created here, `env://`, as src/main.py:31).
    `device` defaults to cuda:<rank>; `model_factory(args, embedding_matrix, n_cat, n_sub)` defaults to `build_model`.
    Returns (model, losses) -- `losses` is a CPU tensor with one entry per step."""
    is_distributed = rank is not None
    rank = rank or 0
    device = _resolve_device(rank, is_distributed, device)
    on_gpu = device.type == "cuda"
    if on_gpu:
        torch.cuda.set_device(device)
    world = 1
    if is_distributed:
        _, world = parallel.init_distributed(rank, getattr(args, "nGPU", None), device=device)       # src/main.py:31
    model = (model_factory or build_model)(args, embedding_matrix, len(category_dict or {}), len(subcategory_dict or {}))
    if getattr(args, "load_ckpt_name", None):
        model.load_state_dict(load_checkpoint(os.path.join(args.model_dir, args.load_ckpt_name))["model_state_dict"])
    model = model.to(device)                                   # before the optimizer: its state follows the parameters' device
    if getattr(args, "deterministic", False) and on_gpu:       # bit-reproducible gradients (fixed-point integer atomics)
        # (only trainable outputs are ever registered with the fixed-point scratch: a frozen 575 M-value title table is not)
        ops.set_deterministic(True, elements=sum(p.numel() for p in model.parameters() if p.requires_grad) + (1 << 20), device=device)
    mode = getattr(args, "dp_mode", None) or ("flat" if on_gpu else "ddp")
    net, bucket, optimizer = model, None, None
    if mode == "flat":
        bucket = parallel.FlatBucket(model, lr=args.lr)        # rank-0 broadcast + one all-reduce + fused Adam per step
    elif mode == "ddp":
        if world > 1:
            net = parallel.wrap_ddp(model, device)             # src/main.py:82
        optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, fused=on_gpu)      # src/main.py:76
    else:
        raise ValueError(f"dp_mode must be 'flat' or 'ddp', got {mode!r}")

    data_file = os.path.join(args.train_data_dir, f"behaviors_np{args.npratio}_{rank}.tsv")            # src/main.py:89
    feed_mode = getattr(args, "feed", None) or ("device" if on_gpu else "host")
    if feed_mode == "device":
        feed = DeviceFeed(IndexedTrainShard(data_file, news_index, args), news_combined, args.batch_size, device)
        n_local = len(feed)
    elif feed_mode == "host":
        dataset = DatasetTrain(data_file, news_index, news_combined, args)
        n_local = (_count_lines(data_file) + args.batch_size - 1) // args.batch_size
    else:
        raise ValueError(f"feed must be 'device' or 'host', got {feed_mode!r}")
    n_batches = parallel.agree_on_batches(n_local, device)     # every rank stops together

    losses, step = [], 0
    for ep in range(getattr(args, "start_epoch", 0), args.epochs):
        loss_sum = torch.zeros((), device=device)
        acc_sum = torch.zeros((), device=device)
        net.train()
        if feed_mode == "device":
            feed.start_epoch()
            batches = (feed.batch(i) for i in range(n_batches))
        else:
            batches = itertools.islice(iter(DataLoader(dataset, batch_size=args.batch_size)), n_batches)
        for cnt, (history, history_mask, candidate, label) in enumerate(batches):
            if feed_mode == "host":
                history = history.to(device, non_blocking=True)
                history_mask = history_mask.to(device, non_blocking=True)
                candidate = candidate.to(device, non_blocking=True)
                label = label.to(device, non_blocking=True)
            bz_loss, y_hat = net(history, history_mask, candidate, label)
            if optimizer is not None:
                optimizer.zero_grad()
            bz_loss.backward()
            if bucket is not None:
                bucket.step()
            else:
                optimizer.step()
            with torch.no_grad():                              # accumulated on the device: no host sync per step
                loss_sum += bz_loss.detach()
                acc_sum += acc(label, y_hat)
                losses.append(bz_loss.detach())
            step += 1
            if cnt % args.log_steps == 0:
                log("[{}][{}] Ed: {}, train_loss: {:.5f}, acc: {:.5f}".format(
                    ep, rank, cnt * args.batch_size, float(loss_sum) / (cnt + 1), float(acc_sum) / (cnt + 1)))
            if max_steps is not None and step >= max_steps:
                break
        if rank == 0 and getattr(args, "model_dir", None):
            os.makedirs(args.model_dir, exist_ok=True)
            torch.save(checkpoint_dict(net, category_dict, subcategory_dict), os.path.join(args.model_dir, f"epoch-{ep + 1}.pt"))
        if max_steps is not None and step >= max_steps:
            break
    return model, (torch.stack(losses).float().cpu() if losses else torch.zeros(0))


@torch.no_grad()
def encode_news(model, news_combined, batch_size, device, shard_over_ranks=False):
    """Full-corpus encode (src/main.py:185-198); the [N+1, news_dim] table stays on the device.  With
    `shard_over_ranks` every rank encodes a contiguous 1/world slice and the slices are all-gathered --
    the reference encodes the whole corpus on every rank."""
    # (a tensor that already lives on the device is taken as it is: the slices below are views, `.to(device)` a no-op)
    ids = news_combined.to(torch.int32) if torch.is_tensor(news_combined) else torch.as_tensor(np.asarray(news_combined), dtype=torch.int32)
    n = ids.shape[0]
    batch_size = max(int(batch_size), 16384)                   # the encoder is row-wise: bigger chunks, same vectors, fewer launches
    world = parallel.world_size() if shard_over_ranks else 1
    rank = dist.get_rank() if world > 1 else 0
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    out = [model.news_encoder(ids[i:min(hi, i + batch_size)].to(device)) for i in range(lo, hi, batch_size)]
    dim = out[0].shape[1] if out else model.args.news_dim
    mine = torch.cat(out, dim=0) if out else torch.zeros(0, dim, device=device)
    if world == 1:
        return mine
    padded = torch.zeros(per, dim, dtype=mine.dtype, device=device)
    padded[: mine.shape[0]] = mine
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)                             # [per, news_dim] fp32 per rank (20 MB at 100k news / 8 ranks)
    return torch.cat(parts, dim=0)[:n]


def _short_history_split(shard, H):
    """[(impressions whose first H - 32 history slots are all masked, H - 32), (the others, 0)] as index arrays; part of the
    shard's preparation (numpy over its mask array), kept on the shard object."""
    cached = getattr(shard, "_nr_short_split", None)
    if cached is None or cached[0] != H:
        m = shard.mask
        m = m.detach().cpu().numpy() if torch.is_tensor(m) else np.asarray(m)
        short = ~(m[:, :H - 32] != 0).any(axis=1)
        cached = (H, [(np.nonzero(short)[0].astype(np.int64), H - 32), (np.nonzero(~short)[0].astype(np.int64), 0)], {})
        try:
            shard._nr_short_split = cached
        except AttributeError:
            pass
    return cached[1], cached[2]


@torch.no_grad()
def score_shard(model, news_vecs, shard: IndexedTestShard, batch_size, device):
    """Rows a13 + f2 on the device: user vectors of every impression of the shard (history vectors gathered from the
    news-vector table), candidate scores (src/main.py:253) and the per-impression ranking metrics.
    Returns (scores [n_cand] device fp32, sums device fp64 [5] = scored count + 4 metric sums)."""
    n = len(shard)
    hist = torch.as_tensor(shard.hist, device=device)
    mask = torch.as_tensor(shard.mask, device=device)
    user = torch.empty(n, news_vecs.shape[1], dtype=torch.float32, device=device)
    # masked user encoder (src/demo.sh:26): the gather can hand over the compute dtype directly (gather + cast in one pass)
    margs = getattr(model, "args", None)
    code = ops.dtype_code(getattr(margs, "compute_dtype", "fp32")) if getattr(margs, "user_log_mask", False) else ops.NR_F32
    batch_size = max(int(batch_size), 8192)                    # the user encoder is row-wise: bigger chunks, same vectors
    indexed = getattr(model.user_encoder, "forward_indexed", None)                 # history as indices into the vector table
    H = hist.shape[1] if hist.dim() == 2 else 0
    if indexed is not None and getattr(margs, "user_log_mask", False) and H > 32 and n > 0:
        # A masked slot contributes nothing to the masked user encoder (attention keys with weight 0, pooling weight 0:
        # src/model/model_utils.py:28,51), so it can simply be left out.  Histories are front-padded (src/dataset.py:17-24): a
        # user whose first H - 32 slots are all masked is encoded from the LAST 32 slots alone -- one 32 x 32 attention tile per
        # head through the title-shape kernel (2.5 x the item rate of the 64-row kernel) and 32 instead of H pooling rows.  The
        # split is made on the host from the shard's own mask array (no device synchronisation); same vectors.
        groups, on_device = _short_history_split(shard, H)
        for g, (idx_np, off) in enumerate(groups):
            if len(idx_np) == 0:
                continue
            sel = on_device.get((g, str(device)))
            if sel is None:
                sel = on_device[(g, str(device))] = torch.as_tensor(idx_np, device=device)
            h_g = hist.index_select(0, sel)[:, off:].contiguous()
            m_g = mask.index_select(0, sel)[:, off:].contiguous()
            for a in range(0, len(idx_np), batch_size):
                b = min(len(idx_np), a + batch_size)
                user.index_copy_(0, sel[a:b], indexed(news_vecs, h_g[a:b], m_g[a:b]).float())
    else:
        for a in range(0, n, batch_size):
            b = min(n, a + batch_size)
            if indexed is not None:
                user[a:b] = indexed(news_vecs, hist[a:b], mask[a:b])
            else:
                log_vecs = ops.embed_gather(news_vecs, hist[a:b], code)               # [B, H, news_dim], device gather
                user[a:b] = model.user_encoder(log_vecs, mask[a:b])                   # src/main.py:247
    offsets = torch.as_tensor(shard.offsets, device=device)
    counts = shard.offsets[1:] - shard.offsets[:-1]
    imp_of = torch.repeat_interleave(torch.arange(n, dtype=torch.int32, device=device), torch.as_tensor(counts, device=device).long())
    cand = torch.as_tensor(shard.cand, device=device)
    scores = ops.score_eval(news_vecs, cand, imp_of, user) if cand.numel() else torch.zeros(0, device=device)
    sums = ops.eval_metrics(scores, torch.as_tensor(shard.label, device=device), offsets, max_cand=int(counts.max()) if n else 0)
    return scores, sums


@torch.no_grad()
def test(rank, args, model, news_index, news_combined, log=logging.info, collect_scores=None, device=None, score_fn=None):
    """One rank of the evaluation job (src/main.py:145-277) on `behaviors_{rank}.tsv`.

    Returns (n_samples, means): n_samples = impressions seen by all ranks, means = [AUC, MRR, nDCG@5, nDCG@10].
    Divisor, as the reference: a single process averages over the SCORED impressions (those with both classes,
    src/main.py:250,275); a distributed run divides the reduced sums by ALL impressions seen (src/main.py:269-273
    reduces `local_sample_num`, which counts the skipped ones too).  `args.eval_divide_by_scored=True` uses the scored
    count in both cases.
    `collect_scores`: optional list that receives (labels, scores) numpy pairs of every impression of this rank.
    `score_fn(model, news_vecs, shard, batch_size, device) -> (scores, sums[5])` defaults to `score_shard` (device)."""
    is_distributed = rank is not None
    rank = rank or 0
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if is_distributed:
        parallel.init_distributed(rank, getattr(args, "nGPU", None), device=device)                  # src/main.py:154
    model.eval()
    news_vecs = encode_news(model, news_combined, args.batch_size, device,
                            shard_over_ranks=is_distributed and getattr(args, "shard_encode", True))
    shard = IndexedTestShard(os.path.join(args.test_data_dir, f"behaviors_{rank}.tsv"), news_index, args)
    scores, sums = (score_fn or score_shard)(model, news_vecs, shard, args.batch_size, device)
    if collect_scores is not None:
        s = scores.cpu().numpy()
        for i in range(len(shard)):
            a, b = shard.offsets[i], shard.offsets[i + 1]
            collect_scores.append((shard.label[a:b], s[a:b]))
    sums = [float(x) for x in (sums.cpu().tolist() if torch.is_tensor(sums) else sums)]
    n_samples, n_scored, metric_sums = parallel.reduce_eval_sums(len(shard), sums[0], sums[1:], device)
    by_scored = getattr(args, "eval_divide_by_scored", False) or not is_distributed
    means = np.asarray(metric_sums) / max(n_scored if by_scored else n_samples, 1)
    if rank == 0:
        log("[*] {} samples: {}".format(n_samples, "\t".join("{:0.2f}".format(x * 100) for x in means)))
    return n_samples, means
