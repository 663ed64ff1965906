#!/bin/bash
# phase ablation of the LDS-DMA NT kernel (results are wrong, timing only): 0 = as is, 8 = no epilogue, 2 = no MFMAs, 4 = no operand DMA
for v in ${NT_LIST:-0 8 2 4 10 12}; do NR_NT_ABLATE=$v python3 bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j.get('kernel_ms_per_step',{})
print('NT_ABLATE=$v', j['ms_per_step'], [(n.split(chr(91))[0],v) for n,v in k.items() if 'gemm_nt_dma' in n])
"; done | tee gpurun_out/nt_ablate.txt
