#!/usr/bin/env python3
"""K sweep of the NT GEMM at the QKV shape: separates the per-k-step cost from the fixed per-tile cost
(pipeline fill + epilogue + stores).  python tools/gemm_ksweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from newsrecommendation_amd import ops
from tools.gemm_probe import timeit

def main():
    dev, M = "cuda", int(os.environ.get("M", 844800))
    for N in (1200, 208, 200):
        for K in (288, 576, 1152, 2304):
            if M * K * 2 > 6e9:
                continue
            a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            b = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
            ms = timeit(lambda: ops.gemm_nt(a, b))
            print(f"nt N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
            del a, b

if __name__ == "__main__":
    main()
