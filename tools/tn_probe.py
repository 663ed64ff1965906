#!/usr/bin/env python3
"""Time one TN weight-gradient GEMM shape: python tools/tn_probe.py N K   (env knobs NR_TN3_* apply)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from newsrecommendation_amd import ops
from tools.gemm_probe import timeit

M = int(os.environ.get("M", 844800))
N, K = int(sys.argv[1]), int(sys.argv[2])
dc = (torch.randn(M, N, device="cuda") * 0.1).to(torch.bfloat16)
a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
ms = timeit(lambda: ops.gemm_tn(dc, a))
knobs = {k: v for k, v in os.environ.items() if k.startswith("NR_TN")}
print(f"tn N={N} K={K} {knobs}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
