python -m pytest tests/test_gpu_model_parity.py tests/test_gpu_scale_parity.py tests/test_gpu_properties.py tests/test_gpu_gemm_wreg.py tests/test_gpu_train_loop.py tests/test_gpu_flat_two_ranks.py -q > gpurun_out/ab_tests.txt 2>&1; tail -4 gpurun_out/ab_tests.txt
bash tools/ab_tree.sh | tee gpurun_out/ab_tree.txt
