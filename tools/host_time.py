"""How long does the HOST need to enqueue one training step (Python + ctypes + launches), next to what the GPU needs to run it?
    python tools/host_time.py [--feed device]
Prints the enqueue time per step (loop wall time before the final synchronize, queue never full for short runs) and the
synchronized step time."""
import sys, time, cProfile, pstats, io
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def main():
    feed = "--feed" in sys.argv and sys.argv[sys.argv.index("--feed") + 1] == "device"
    from newsrecommendation_amd import parallel, train as TR
    from newsrecommendation_amd.model import NRMS
    import numpy as np
    dev = torch.device("cuda", 0)
    args = bench.make_args("bf16")
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    table = torch.randn(30000, 300, generator=g) * 0.4
    table[0] = 0
    model = NRMS.Model(args, table.numpy()).to(dev).train()
    fb = parallel.FlatBucket(model, lr=1e-4)
    batches = bench.synth_batches(args, 512, 30000, 4, 100, dev)
    fd = None
    if feed:
        n_news, n_lines = 65000, 512 * 64
        comb = bench.synth_news_table(args, n_news, 30000, 3).numpy()
        rnd = np.random.RandomState(100)
        hl = rnd.randint(0, 51, n_lines)
        hist = rnd.randint(1, n_news + 1, (n_lines, 50)).astype(np.int32)
        mask = (np.arange(50)[None, :] >= (50 - hl)[:, None]).astype(np.float32)
        hist[mask == 0] = 0
        sh = bench.ArrayShard(n_lines, hist=hist, mask=mask, pos=rnd.randint(1, n_news + 1, n_lines).astype(np.int32),
                              neg=rnd.randint(1, n_news + 1, (n_lines, 4)).astype(np.int32), npratio=4)
        fd = TR.DeviceFeed(sh, comb, 512, dev)
        fd.start_epoch()
    def step(i):
        b = fd.batch(i) if fd is not None else batches[i % 4]
        loss, _ = model(*b)
        loss.backward()
        fb.step()
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for i in range(N):
        step(6 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"feed={'device' if feed else 'resident'}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, synchronized {1e3 * (t2 - t0) / N:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(10):
        step(40 + i)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print("\n".join(s.getvalue().splitlines()[:40]))

main()
