#!/usr/bin/env python3
"""Phase ablation of the LDS-DMA NT GEMM (NR_NT_ABLATE bits: 1 no output stores, 2 no MFMAs, 4 no operand DMA, 8 no epilogue).
Measurement only -- the ablated launches produce wrong results.  python tools/nt_ablate.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from newsrecommendation_amd import ops, _lib
from gemm_probe import timeit

def main():
    _lib.set_option("NT_WREG", 0)      # these tools take the TILED LDS-DMA kernel apart (the default for these shapes is gemm_nt_wreg_kernel)
    M = int(os.environ.get("M", 253440))
    for name, N, K in [("qkv", 1200, 304), ("fc1", 200, 400), ("poolbwd-like", 400, 200), ("dx-like", 304, 1200)]:
        a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        Kr = (K + 31) // 32 * 32
        bfull = torch.zeros(N, Kr, device="cuda", dtype=torch.bfloat16)
        bfull[:, :K] = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
        b = bfull[:, :K]
        out = []
        for abl in (0, 1, 8, 2, 4, 6, 7, 14, 15):
            _lib.set_option("NT_ABLATE", abl)
            out.append(f"{abl}:{timeit(lambda: ops.gemm_nt(a, b)):.3f}")
        _lib.set_option("NT_ABLATE", 0)
        print(f"{name:12s} M={M} N={N} K={K}  ms by ablation mask  " + "  ".join(out), flush=True)

if __name__ == "__main__":
    main()
