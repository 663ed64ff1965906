#!/bin/bash
# sweep an integer library option over a few values and print the kernels whose label contains a pattern
OPT=$1; PAT=$2; shift 2
for v in "$@"; do
  env NR_$OPT=$v python3 bench.py --no-also --no-cpu-baseline --steps 15 --warmup 5 --all-kernels 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$OPT=$v', j['ms_per_step'], [(n.split('[')[0],v) for n,v in k.items() if '$PAT' in n])"
done
