#!/bin/bash
# where does the fused pooling forward spend its time?  (results of the ablated runs are wrong by design)
for ab in 0 1 2 3 4 7 8 16 24 31; do
  NR_POOL_ABLATE=$ab python3 bench.py --no-also --no-cpu-baseline --steps 10 --warmup 4 --all-kernels 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('ablate=$ab', [ (n.split('[')[0],v) for n,v in k.items() if n.startswith('pool_fused')])"
done
