"""Data-parallel plumbing.  One process per GPU; `torch.distributed` backend 'nccl' is RCCL
on ROCm (xGMI inside a node), 'gloo' on CPU for the host-logic tests.

The reference wraps the model in DistributedDataParallel (src/main.py:82): rank-0 parameter broadcast at
construction, gradient all-reduce-mean in backward.  `wrap_ddp` does exactly that; `FlatBucketDP` is the
single-collective alternative: all gradients live in ONE contiguous fp32 bucket (4 MB for NRMS, + the table
gradient when it is trainable) and are averaged with one all_reduce per step — on a fully connected 8-GPU xGMI
mesh that message is latency bound, so one launch beats DDP's several buckets.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Process-group init from the torchrun environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def wrap_ddp(model, local_rank):
    """Same wrapper and defaults as src/main.py:82."""
    ids = [local_rank] if next(model.parameters()).is_cuda else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids)


class FlatBucketDP:
    """Gradient averaging through one flat bucket.

        dp = FlatBucketDP(model)         # broadcasts rank 0's parameters (DDP construction semantics)
        loss.backward(); dp.allreduce_grads(); optimizer.step()
    """

    def __init__(self, model, group=None, broadcast=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.bucket = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:                      # grads become views of the bucket: no per-step flatten copy
            p.grad = self.bucket[off:off + p.numel()].view_as(p)
            off += p.numel()
        if broadcast and self.world > 1:
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
            dist.broadcast(flat, src=0, group=group)
            off = 0
            for p in model.parameters():
                p.data.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def zero_grad(self):
        self.bucket.zero_()

    def allreduce_grads(self):
        off = 0
        for p in self.params:                      # autograd may have replaced .grad: fold it back into the bucket
            view = self.bucket[off:off + p.numel()].view_as(p)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
            off += p.numel()
        if self.world > 1:
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            self.bucket.div_(self.world)
