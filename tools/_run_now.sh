python -m pytest tests/test_gpu_ops_golden.py tests/test_gpu_model_parity.py tests/test_gpu_train_loop.py tests/test_gpu_properties.py -q > gpurun_out/ab_tests.txt 2>&1; tail -4 gpurun_out/ab_tests.txt
for i in 1 2; do for D in ab_tree .; do (cd $D && python3 bench.py --eval --no-also --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=j.get('kernel_ms_per_step',{})
print('EVAL $D', j['ms_per_step'], ' '.join(f'{n.split(chr(91))[0]}={v}' for n,v in list(k.items())[:5]))
"); done; done | tee gpurun_out/ab_eval.txt
