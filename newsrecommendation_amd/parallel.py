"""Data-parallel plumbing of the train / eval loops.  One process per GPU;
`torch.distributed` backend 'nccl' is RCCL on ROCm (xGMI inside a node), 'gloo' on CPU for the host-logic tests.

The reference wraps the model in DistributedDataParallel (src/main.py:82): rank-0 parameter broadcast at
construction, gradient all-reduce-mean in backward, then `optim.Adam.step()` (src/main.py:76,110).  Two modes here,
chosen by `args.dp_mode` in `train.train()`:

  "flat" (default on the GPU)  `FlatBucket`: every trainable parameter lives in ONE contiguous fp32 buffer, every
        gradient in another, Adam's moments in two more.  The autograd functions of `ops.py` accumulate straight
        into the gradient buffer, `step()` is ONE all-reduce over it (4 MB for NRMS + the word-table gradient when
        it is trainable -- latency bound on the xGMI mesh, so one message beats DDP's several buckets) and ONE HIP
        kernel (`nr_adam_step`) that applies the 1/world average, Adam's update and the next step's zero_grad.
  "ddp"   the reference's own objects: `DistributedDataParallel` + `torch.optim.Adam` (fused on the GPU).

Both have the reference's semantics (same broadcast, same mean, same update rule).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(rank=None, world=None, backend=None, device=None):
    """Process-group init, `env://` as src/main.py:31,154.  rank / world default to the torchrun environment
    (RANK / WORLD_SIZE); MASTER_ADDR / MASTER_PORT default to 127.0.0.1:8888 (src/main.py:286-287 uses localhost:8888).
    Returns (rank, world).  world == 1: no process group is created."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "8888")
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device(device)
        dist.init_process_group(backend, init_method="env://", world_size=world, rank=rank, **kw)
    return rank, world


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def agree_on_batches(n_local_batches, device, group=None):
    """Number of batches EVERY rank can run.  Shards written round-robin (`i % nGPU`, src/prepare_data.py:39-41) differ by
    up to one sample, so ranks can disagree on the batch count and the extra gradient all-reduce of the longer ranks
    never completes (the reference's latent hang).  One MIN all-reduce before the epoch fixes the count."""
    if world_size(group) == 1:
        return int(n_local_batches)
    t = torch.tensor([int(n_local_batches)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def wrap_ddp(model, device):
    """Same wrapper and defaults as src/main.py:82."""
    ids = [torch.device(device).index] if torch.device(device).type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids)


def reduce_eval_sums(n_samples, n_scored, metric_sums, device, group=None):
    """src/main.py:269-273: SUM-reduce of the sample count and the four metric sums to rank 0, as ONE message
    [n_samples, n_scored, sum AUC, sum MRR, sum nDCG@5, sum nDCG@10] (fp64).  Every rank gets the tuple back; only
    rank 0's is the global one."""
    t = torch.tensor([float(n_samples), float(n_scored), *[float(x) for x in metric_sums]], dtype=torch.float64, device=device)
    if world_size(group) > 1:
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=group)
    v = t.cpu().tolist()
    return int(round(v[0])), int(round(v[1])), v[2:]


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatBucket:
    """Parameters, gradients and Adam state of `model` as four flat fp32 buffers.

        fb = FlatBucket(model, lr=args.lr)      # model already on its device; broadcasts rank 0's parameters
        loss.backward()                         # ops.py accumulates into fb.grad directly
        fb.step()                               # all-reduce (world > 1) + fused Adam + zero_grad, two launches

    Layout: the three projections of every MultiHeadSelfAttention are adjacent (weights [3N, d_model], then biases [3N])
    so the QKV GEMMs read / write them without a concatenation; every other parameter starts on a 256-byte boundary.
    Frozen parameters stay where they are."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, group=None, broadcast=True):
        from .model.model_utils import MultiHeadSelfAttention
        self.group, self.world = group, world_size(group)
        self.lr, self.betas, self.eps, self.t = float(lr), (float(betas[0]), float(betas[1])), float(eps), 0
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("FlatBucket: the model has no trainable parameter")
        dev = params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in params):
            raise ValueError("FlatBucket: trainable parameters must be fp32 on one device")
        # segments: parameters laid out back to back, each segment starting on a 256-byte boundary
        segments, seen, groups = [], set(), []
        for m in model.modules():
            if isinstance(m, MultiHeadSelfAttention):
                ws, bs = [m.W_Q.weight, m.W_K.weight, m.W_V.weight], [m.W_Q.bias, m.W_K.bias, m.W_V.bias]
                if all(p.requires_grad and id(p) not in seen for p in ws + bs):
                    groups.append((m, len(segments)))        # weights segment, then biases segment
                    segments += [ws, bs]
                    seen.update(id(p) for p in ws + bs)
        rest = [p for p in params if id(p) not in seen]
        # the largest parameter (the word-embedding table: 9 of NRMS's 10 M values) goes LAST: its gradient is all-reduced on
        # its own as soon as the backward has produced it (see _early_allreduce), everything in front of it in one more call
        big = max(rest, key=lambda p: p.numel()) if rest else None
        if big is not None and big.numel() >= (1 << 20):
            rest = [p for p in rest if p is not big] + [big]
        else:
            big = None
        segments += [[p] for p in rest]
        order, offs, seg_off, off = [], [], [], 0
        for seg in segments:
            off = _align(off)
            seg_off.append(off)
            for p in seg:
                order.append(p)
                offs.append(off)
                off += p.numel()
        self.numel = _align(off)
        self.param = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.param)
        self.exp_avg = torch.zeros_like(self.param)
        self.exp_avg_sq = torch.zeros_like(self.param)
        self.params, self.views = order, []
        with torch.no_grad():
            for p, o in zip(order, offs):
                pv, gv = self.param[o:o + p.numel()].view_as(p), self.grad[o:o + p.numel()].view_as(p)
                pv.copy_(p)
                p.data = pv                          # the Parameter object (and its state_dict key) stays, its storage moves
                p.grad = gv
                p._nr_grad = gv
                self.views.append((pv, gv))
        for m, si in groups:
            N, d_model = segments[si][0].shape
            o_w, o_b = seg_off[si], seg_off[si + 1]
            m._nr_flat = {"w": self.param[o_w:o_w + 3 * N * d_model].view(3 * N, d_model), "b": self.param[o_b:o_b + 3 * N],
                          "gw": self.grad[o_w:o_w + 3 * N * d_model].view(3 * N, d_model), "gb": self.grad[o_b:o_b + 3 * N]}
        self._model = model
        from . import ops
        # the small weights' packed copies are refreshed in one launch per step; their stamp follows the version counter of
        # EVERY parameter of the bucket (`p.data = view` leaves each Parameter its own counter: a torch-side in-place write such
        # as load_state_dict bumps that one, not the flat buffer's)
        ops.pack_cache.register_buffer(self.param, versions=self._versions)
        self._early_work, self._big_off, self.early_calls = None, None, 0
        if big is not None and self.world > 1:
            # The early all-reduce splits the step's collective in two; whether the backward fires it depends on per-process
            # state (deterministic mode, a trainable table).  Ranks that disagreed would issue collectives of different sizes,
            # so they agree ONCE, here: one MIN all-reduce of "I would fire it".
            mine = torch.tensor([1 if ops._det_scratch is None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(mine, op=dist.ReduceOp.MIN, group=group)
            if int(mine.item()) == 1:
                self._big_off = offs[[id(q) for q in order].index(id(big))]
                big._nr_grad_ready = self._early_allreduce   # ops.MHSAFunction.backward calls it once the table gradient is complete
        if broadcast and self.world > 1:             # DDP construction semantics (src/main.py:82): rank 0's parameters win
            dist.broadcast(self.param, src=0, group=group)
        self._params_changed()

    def _params_changed(self, repacked=()):
        """The kernels wrote the parameters behind autograd's back (no version bump): packed compute-dtype copies of the
        bucket's OWN tables must be rebuilt -- except those the Adam kernel rewrote itself (`repacked`: parameter ids).
        Frozen tables (e.g. NAML's 2.3 GB title table) are not touched."""
        from . import ops
        for p in self.params:
            if id(p) not in repacked:
                ops.table_cache.invalidate(p)
        ops.bump_param_epoch()

    def _pack_jobs(self):
        """The bucket's embedding tables that have an up-to-date packed bf16 copy (ops.table_cache): nr_adam_step_packed writes
        the new values into that copy in the same pass, so the next forward finds it current (no nr_cast_pad of the table)."""
        from . import _lib, ops
        jobs, ids = [], set()
        for p in self.params:
            ents = ops.table_cache.current(p)
            if not ents or len(jobs) + len(ents) > _lib.ADAM_PACK_MAX:
                continue
            first = (p.data_ptr() - self.param.data_ptr()) // 4
            mine = []
            for code, cols, packed in ents:
                ok = (code == _lib.NR_BF16 and cols % 4 == 0 and first % 4 == 0 and p.numel() % cols == 0 and p.numel() < 2 ** 32
                      and packed.dtype == torch.bfloat16 and packed.is_contiguous() and packed.dim() == 2
                      and packed.shape[0] == p.numel() // cols and packed.shape[1] >= cols and packed.shape[1] % 4 == 0)
                if not ok:
                    mine = None
                    break
                mine.append(_lib.PackJob(first, p.numel(), cols, packed.shape[1], packed.data_ptr()))
            if mine:
                jobs += mine
                ids.add(id(p))
        return jobs, ids

    def _versions(self):
        return tuple(p._version for p in self.params)

    def zero_grad(self):
        """Also the abort path of a step: an early table all-reduce that is still in flight is waited for and dropped, so
        the next backward starts clean."""
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None
        self.grad.zero_()

    def _early_allreduce(self):
        """Called from the backward pass right after the table-gradient kernel was enqueued (one backward per step): its
        all-reduce starts now, on the collective's own stream, underneath the rest of the backward."""
        if self.world > 1:
            if self._early_work is not None:
                raise RuntimeError("FlatBucket: a second backward pass before step() -- the table gradient of the first one is "
                                   "already being all-reduced (one backward per optimizer step, as src/main.py:104-110)")
            self._early_work = dist.all_reduce(self.grad[self._big_off:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.early_calls += 1

    def allreduce(self):
        """The collective(s) of a step: SUM over ranks (the 1/world of the mean is applied inside the Adam kernel).  One call
        over the whole flat gradient -- or, when the backward announced the table gradient early, one over what lies in front
        of it plus the wait for the early one."""
        if self.world > 1:
            if self._early_work is not None:
                if self._big_off > 0:
                    dist.all_reduce(self.grad[:self._big_off], op=dist.ReduceOp.SUM, group=self.group)
                self._early_work.wait()
                self._early_work = None
            else:
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)

    def adam_step(self, zero_grad=True):
        if not self.param.is_cuda:
            raise RuntimeError("FlatBucket.adam_step runs the HIP kernel nr_adam_step: parameters must live on the GPU "
                               "(libnrhip has no CPU fallback)")
        from . import _lib
        self.t += 1
        jobs, repacked = self._pack_jobs()
        arr = (_lib.PackJob * len(jobs))(*jobs) if jobs else None
        _lib.check(_lib.lib().nr_adam_step_packed(self.param.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                                  self.exp_avg_sq.data_ptr(), self.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                                                  self.t, 1.0 / self.world, int(zero_grad), arr, len(jobs),
                                                  torch.cuda.current_stream().cuda_stream),
                   "nr_adam_step_packed")
        self._params_changed(repacked)

    def step(self):
        self.allreduce()
        self.adam_step(zero_grad=True)

    # -- checkpoint interop: torch.optim.Adam's state_dict layout (per-parameter exp_avg / exp_avg_sq / step)
    def state_dict(self):
        names = {id(p): n for n, p in self._model.named_parameters()}
        st = {}
        for p, (pv, gv) in zip(self.params, self.views):
            o = pv.data_ptr() - self.param.data_ptr()
            o //= 4
            st[names[id(p)]] = {"step": self.t, "exp_avg": self.exp_avg[o:o + p.numel()].view_as(p).clone(),
                                "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view_as(p).clone()}
        return {"state": st, "lr": self.lr, "betas": self.betas, "eps": self.eps}
