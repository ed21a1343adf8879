#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/q_tests.log 2>&1; tail -5 gpurun_out/q_tests.log
OFA_GEMM_TILE=0 timeout 300 python tools/gemm_tile_sweep.py 13312 2048 > gpurun_out/q_sweep_new.txt 2>&1; grep -v amdgpu gpurun_out/q_sweep_new.txt
OFA_GEMM_TILE=22 OFA_GEMM_SPLIT_MIN_K=1000000 timeout 300 python tools/gemm_tile_sweep.py 13312 > gpurun_out/q_sweep_new22.txt 2>&1; grep -v amdgpu gpurun_out/q_sweep_new22.txt
timeout 600 python bench.py --steps 30 --warmup 8 > gpurun_out/q_bench.log 2>&1; tail -1 gpurun_out/q_bench.log | cut -c1-400
