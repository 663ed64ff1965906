#!/usr/bin/env python3
"""Which torch (non-libnrhip) GPU kernels does one NRMS train step launch, and from where?  python tools/step_ops.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from newsrecommendation_amd import parallel
from newsrecommendation_amd.model import NRMS

def main():
    dev = torch.device("cuda", 0)
    args = bench.make_args("bf16")
    args.freeze_embedding = False; args.compact_history = False
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    table = torch.randn(30000, args.word_embedding_dim, generator=g) * 0.4
    table[0] = 0
    model = NRMS.Model(args, table.numpy()).to(dev).train()
    bucket = parallel.FlatBucket(model, lr=1e-4)
    batches = bench.synth_batches(args, 512, 30000, 4, 100, dev, dense=False)
    def step(i):
        hist, mask, cand, label = batches[i % 4]
        loss, _ = model(hist, mask, cand, label)
        loss.backward()
        bucket.step()
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
        if e.device_time_total > 0 and e.key.startswith("aten::"):
            rows.append((e.device_time_total / 3, e.count / 3, e.key, str(e.input_shapes)[:90], [s for s in e.stack if "newsrecommendation_amd" in s or "bench" in s][:3]))
    rows.sort(reverse=True)
    for us, n, k, shp, st in rows[:40]:
        print(f"{us:8.1f} us/step  x{n:4.1f}  {k:28s} {shp}")
        for s in st:
            print("            ", s[-110:])

if __name__ == "__main__":
    main()
