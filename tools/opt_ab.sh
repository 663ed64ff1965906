#!/bin/bash
# same-box A/B of one library option:  bash tools/opt_ab.sh NR_ATTN_BWD_OCC4 [bench args]   (alternates unset / =1, three rounds)
O=$1; shift
for i in 1 2 3; do for v in "" 1; do
  env ${v:+$O=$v} python3 bench.py --no-also --no-cpu-baseline --steps 30 --warmup 6 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']
print('$O=${v:-unset}', j['ms_per_step'], [(n.split(chr(91))[0],x) for n,x in list(k.items())[:3]])"
done; done
