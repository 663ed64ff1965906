#!/usr/bin/env python3
"""Kernel-level GEMM timing on the GPU box: python tools/gemm_probe.py  (uses the nr_gemm_* hooks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from newsrecommendation_amd import ops, _lib

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def main():
    dev = "cuda"
    M = int(os.environ.get("M", 844800))
    shapes = [("qkv", 1200, 304), ("fc1", 200, 400), ("poolbwd", 400, 200), ("dx", 304, 1200)]
    for name, N, K in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        Kr = (K + 31) // 32 * 32
        bfull = torch.zeros(N, Kr, device=dev, dtype=torch.bfloat16)
        bfull[:, :K] = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
        b = bfull[:, :K]          # row stride Kr, zero padded: eligible for the LDS-DMA kernel
        ms = timeit(lambda: ops.gemm_nt(a, b))
        print(f"nt {name:8s} M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s  out {M*N*2/ms/1e6:.0f} GB/s")
        ref = (a[:4096].float() @ b.float().t())
        got = ops.gemm_nt(a[:4096].contiguous(), b).float()
        print("   max err", float((ref - got).abs().max()), "ref max", float(ref.abs().max()))
    for name, N, K in [("dWqkv", 1200, 304), ("dW1", 200, 400)]:
        dc = (torch.randn(M, N, device=dev) * 0.1).to(torch.bfloat16)
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        ms = timeit(lambda: ops.gemm_tn(dc, a))
        print(f"tn {name:8s} M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s  in {M*(N+K)*2/ms/1e6:.0f} GB/s")

if __name__ == "__main__":
    main()
