#!/bin/bash
# Same-box A/B of library builds:  bash tools/ab.sh newsrecommendation_amd/ab/libnrhip_A.so newsrecommendation_amd/ab/libnrhip_B.so [bench args]
# Alternates A B A B (boxes and clocks drift); prints ms/step and the top kernels of every run.
A=$1; B=$2; shift 2
for i in 1 2; do
  for L in "$A" "$B"; do
    NRHIP_LIB=$(realpath "$L") python3 bench.py --no-also --no-cpu-baseline --steps 30 --warmup 6 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j.get('kernel_ms_per_step',{})
print('$L', j['ms_per_step'], ' '.join(f'{n.split(chr(91))[0]}={v}' for n,v in list(k.items())[:9]))
"
  done
done
