#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from newsrecommendation_amd import ops
M = 844800
name = sys.argv[1] if len(sys.argv) > 1 else "dx"
N, K = {"qkv": (1200, 304), "dx": (304, 1200), "fc1": (200, 400)}[name]
a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
Kr = (K + 31) // 32 * 32
bfull = torch.zeros(N, Kr, device="cuda", dtype=torch.bfloat16)
bfull[:, :K] = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
b = bfull[:, :K]
for _ in range(3):
    ops.gemm_nt(a, b)
torch.cuda.synchronize()
