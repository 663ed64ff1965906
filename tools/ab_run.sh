#!/bin/bash
# one GPU call: chosen GPU tests, then an A/B of ab/libnrhip_A.so (saved baseline) against the fresh build
python -m pytest $AB_TESTS -q -x > gpurun_out/ab_tests.txt 2>&1; tail -4 gpurun_out/ab_tests.txt
bash tools/ab.sh newsrecommendation_amd/ab/libnrhip_A.so newsrecommendation_amd/libnrhip.so 2>&1 | tee gpurun_out/ab_bench.txt
