python -m pytest tests/test_gpu_scale_parity.py -q -x -k "deterministic" > gpurun_out/r03j_det.txt 2>&1; tail -3 gpurun_out/r03j_det.txt
for v in 1 0 1 0; do NR_TN3_ATOMIC=$v python3 bench.py --no-also --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j.get('kernel_ms_per_step',{})
print('TN3_ATOMIC=$v', j['ms_per_step'], [(n.split(chr(91))[0],v) for n,v in k.items() if 'tn3' in n])
"; done | tee gpurun_out/r03j_ab.txt
