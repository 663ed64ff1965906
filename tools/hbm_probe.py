#!/usr/bin/env python3
"""HBM write / copy rates seen by plain torch kernels on this box (context for the GEMM epilogue numbers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_probe import timeit

def main():
    for gb in (0.5, 2.0):
        n = int(gb * 1e9 / 2)
        x = torch.empty(n, device="cuda", dtype=torch.bfloat16)
        y = torch.empty(n, device="cuda", dtype=torch.bfloat16)
        ms = timeit(lambda: x.fill_(1.0))
        print(f"fill {gb} GB: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s written")
        ms = timeit(lambda: y.copy_(x))
        print(f"copy {gb} GB: {ms:.3f} ms  {2*gb/ms*1e3:.0f} GB/s read+write")
        ms = timeit(lambda: x.sum())
        print(f"sum  {gb} GB: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s read")

if __name__ == "__main__":
    main()
