#!/usr/bin/env python3
"""Phase timeline of the LDS-DMA NT GEMM from in-kernel s_memrealtime stamps (NR_NT_ABLATE bit 64; measurement only).
python tools/nt_trace.py [N K]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from newsrecommendation_amd import ops, _lib

def main():
    _lib.set_option("NT_WREG", 0)      # these tools take the TILED LDS-DMA kernel apart (the default for these shapes is gemm_nt_wreg_kernel)
    M = int(os.environ.get("M", 253440))
    N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1200, 304)
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    Kr = (K + 31) // 32 * 32
    bfull = torch.zeros(N, Kr, device="cuda", dtype=torch.bfloat16)
    bfull[:, :K] = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
    b = bfull[:, :K]
    for _ in range(3):
        ops.gemm_nt(a, b)
    _lib.set_option("NT_ABLATE", 64 | int(os.environ.get("ABL", 0)))
    ops.gemm_nt(a, b)
    torch.cuda.synchronize()
    _lib.set_option("NT_ABLATE", 0)
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    assert _lib.lib().nr_debug_nt_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(4096, 8)
    t = t[t[:, 0] > 0]
    st = t[:, :7].astype(np.int64)
    hw = t[:, 7].astype(np.int64)
    # HW_ID (gfx9): wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ... ; xcc in XCC_ID register (not here)
    cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    d = np.diff(st, axis=1) * 10.0 / 1000.0      # us (100 MHz)
    names = ["entry->setup done", "first stage wait", "main loop", "epi LDS pass0", "store0 + LDS pass1", "store1 + drain"]
    print(f"ABL={os.environ.get('ABL', 0)} N={N} K={K} M={M}: {len(t)} workgroups traced; per-phase microseconds (median / p10 / p90)")
    for i, nme in enumerate(names):
        c = d[:, i]
        print(f"  {nme:22s} {np.median(c):7.2f} {np.percentile(c,10):7.2f} {np.percentile(c,90):7.2f}")
    tot = (st[:, 6] - st[:, 0]) / 100.0
    print(f"  whole workgroup        {np.median(tot):7.2f} {np.percentile(tot,10):7.2f} {np.percentile(tot,90):7.2f}")
    # gaps between consecutive workgroups on the same (se, sh, cu) slot, by start time -- only meaningful within one XCD; blockIdx & 7 = XCD
    xcd = np.arange(4096)[: len(t)] & 7
    gaps = []
    for x in range(8):
        for key in set(zip(se[xcd == x], sh[xcd == x], cu[xcd == x])):
            sel = (xcd == x) & (se == key[0]) & (sh == key[1]) & (cu == key[2])
            s = st[sel]; o = np.argsort(s[:, 0]); s = s[o]
            gaps += list((s[1:, 0] - s[:-1, 6]) / 100.0)
    gaps = np.array(gaps)
    if len(gaps):
        print(f"  gap end->next start on a CU: median {np.median(gaps):.2f} us, p10 {np.percentile(gaps,10):.2f}, p90 {np.percentile(gaps,90):.2f} ({len(gaps)} pairs)")
    span = (st[:, 6].max() - st[:, 0].min()) / 100.0
    print(f"  span of traced workgroups {span:.1f} us")

if __name__ == "__main__":
    main()
