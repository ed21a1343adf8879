#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in 0 22 34 44; do OFA_GEMM_TILE=$t python tools/gemm_tile_sweep.py 13312 12800 14336 2>&1 | grep -v amdgpu.ids; done > gpurun_out/e_tile_sweep.txt
wc -l gpurun_out/e_tile_sweep.txt
