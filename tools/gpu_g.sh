#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in 0 42; do OFA_SWEEP_CHECK=1 OFA_GEMM_TILE=$t timeout 300 python tools/gemm_tile_sweep.py 13312 2>&1 | grep -v amdgpu.ids; done > gpurun_out/g_tile_sweep.txt
OFA_GEMM_PERSIST=0 OFA_SWEEP_CHECK=1 OFA_GEMM_TILE=42 timeout 300 python tools/gemm_tile_sweep.py 13312 2>&1 | grep -v amdgpu.ids | sed 's/tile42/tile42np/' >> gpurun_out/g_tile_sweep.txt
cat gpurun_out/g_tile_sweep.txt | grep -v TN
OFA_GEMM_TILE=42 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -3
OFA_GEMM_TILE=42 timeout 600 python tools/gemm_sweep.py 2>&1 | tail -3
