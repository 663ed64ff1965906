python -m pytest tests/test_gpu_feed_metrics_adam.py tests/test_gpu_model_parity.py tests/test_gpu_scale_parity.py tests/test_gpu_properties.py tests/test_gpu_train_loop.py tests/test_gpu_formats.py tests/test_gpu_flat_two_ranks.py -q > gpurun_out/ab_tests.txt 2>&1; tail -5 gpurun_out/ab_tests.txt
for i in 1 2 3; do python3 bench.py --no-also --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', j['ms_per_step'])"; done
python3 bench.py --no-also --no-cpu-baseline --model NAML --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NAML ms/step', j['ms_per_step'])"
