#!/bin/bash
# same-box A/B of two whole trees (python + library): ab_tree/ (an export of HEAD with its built library) against the working tree
for i in 1 2 3; do
  for D in ab_tree .; do
    (cd $D && python3 bench.py --no-also --no-cpu-baseline --steps 30 --warmup 6 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$D', j['ms_per_step'])")
  done
done
