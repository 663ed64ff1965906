#!/bin/bash
# eval-workload A/B of two library builds: bash tools/ab_eval.sh A.so B.so
for i in 1 2; do for L in "$1" "$2"; do
  NRHIP_LIB=$(realpath "$L") python3 bench.py --eval --no-also --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j.get('kernel_ms_per_step',{})
print('EVAL $L', j['ms_per_step'], ' '.join(f'{n.split(chr(91))[0]}={v}' for n,v in list(k.items())[:6]))
"; done; done
