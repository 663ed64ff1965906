#!/bin/bash
# kernel sequence of ONE training step (between two adam launches), with durations: bash tools/step_trace.sh > gpurun_out/step_trace.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o st -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-also --no-prof > /tmp/st.log 2>&1 || echo trace failed
f=$(find /tmp/st -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['End_Timestamp'])
prev=t0
tot=0
for r in rows[a+1:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name']
    n=n.replace('(anonymous namespace)::','').replace('void ','')
    print(f"{(s-t0)/1e3:9.1f} gap {(s-prev)/1e3:6.1f} dur {(e-s)/1e3:7.1f}  {n[:100]}")
    prev=max(prev,e); tot+=e-s
print('sum of durations us', tot/1e3, 'span us', (prev-t0)/1e3)
PY
