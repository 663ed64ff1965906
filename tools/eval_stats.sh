#!/bin/bash
# rocprofv3 kernel stats of the eval workload: bash tools/eval_stats.sh > gpurun_out/eval_stats.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev -o ev -- python3 bench.py --eval --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-prof > /tmp/ev.log 2>&1 || echo trace failed
f=$(find /tmp/ev -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/1e6/4:9.3f} ms/pass {int(r['Calls'])/4:8.1f} calls/pass {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:120]}")
PY
