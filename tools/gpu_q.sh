#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_trainstep_gpu.py -x -q -m gpu > gpurun_out/q_tests.log 2>&1; tail -3 gpurun_out/q_tests.log
timeout 600 python bench.py --workload cfg2b --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/q_bench2b.log 2>&1; tail -1 gpurun_out/q_bench2b.log | cut -c1-200
