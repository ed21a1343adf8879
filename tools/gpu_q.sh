#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/q_tests_full.log 2>&1; tail -3 gpurun_out/q_tests_full.log
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/q_bench.log 2>&1; tail -1 gpurun_out/q_bench.log | cut -c1-300
timeout 600 python bench.py --dtype fp16 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/q_bench_fp16.log 2>&1; tail -1 gpurun_out/q_bench_fp16.log | cut -c1-300
