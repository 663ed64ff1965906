"""CPU, world_size 2, gloo: the data-parallel plumbing (flat-bucket gradient averaging with rank-0 broadcast, shard
assignment, common batch count).  The HIP ops need a GPU, so the model here is a plain nn.Module: what is under test
is the collective logic that bench.py / train.py use at N > 1."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from newsrecommendation_amd import data as D
from newsrecommendation_amd.parallel import FlatBucketDP


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different initial weights per rank on purpose
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    dp = FlatBucketDP(model)                            # must broadcast rank 0's weights
    w0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    torch.manual_seed(7 + rank)
    x, y = torch.randn(8, 6), torch.randn(8, 3)         # each rank its own shard
    for _ in range(3):
        dp.zero_grad()
        loss = ((model(x) - y) ** 2).mean()
        loss.backward()
        dp.allreduce_grads()
        opt.step()
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # plain lists, not tensors: a tensor travels as a file descriptor that the parent has to fetch from this process,
    # which races with the worker's exit
    q.put((rank, w0.tolist(), w.tolist(), x.tolist(), y.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_matches_single_process_mean_gradient():
    world, port = 2, 29500 + os.getpid() % 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0a, wa, xa, ya), (_, w0b, wb, xb, yb) = [(r[0],) + tuple(torch.tensor(v) for v in r[1:]) for r in res]
    assert torch.equal(w0a, w0b)                        # rank-0 broadcast at construction
    assert torch.allclose(wa, wb, atol=0, rtol=0)       # identical parameters after 3 averaged steps
    # single-process replay: gradient = mean over the two shards' gradients
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        (0.5 * (((ref(xa) - ya) ** 2).mean() + ((ref(xb) - yb) ** 2).mean())).backward()
        opt.step()
    wr = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    assert torch.allclose(wa, wr, atol=1e-6)


def test_shard_assignment_and_common_batches():
    lines = [f"{i}\tU1\tt\tN1 N2\tN3-1 N4-0 N5-0\n" for i in range(101)]
    shards = D.shard_training_lines(lines, 2, 1, 0)
    assert [len(s) for s in shards] == [51, 50] and set(shards[0]).isdisjoint(shards[1])
    assert D.common_batch_count([len(s) for s in shards], 8) == 6
    t = D.shard_testing_lines(lines, 2)
    assert t[0] == lines[0::2] and t[1] == lines[1::2]
