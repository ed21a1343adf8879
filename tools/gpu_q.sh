#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export OFASYS_AMD_LIB=$GRAFT_REPO_ROOT/tools/experiments/_build/libofasys_amd_deep.so
(
for t in 33 35 22; do OFA_GEMM_TILE=$t OFA_SWEEP_CHECK=1 timeout 300 python tools/gemm_tile_sweep.py 13312; done
) > gpurun_out/q_sweep_deep.txt 2>&1
grep -v amdgpu gpurun_out/q_sweep_deep.txt
