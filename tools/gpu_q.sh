#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export OFASYS_AMD_LIB=$GRAFT_REPO_ROOT/tools/experiments/_build/libofasys_amd_p8.so
export OFA_SWEEP_SHAPES="NT,13312,2304,768;NT,13312,3072,768;NN,13312,3072,768;NN,13312,2304,768;NT,13312,1536,768;NT,12800,3072,768;NT,13312,4096,1024;NT,1000,3072,768;NT,13312,768,3072"
(for t in 85 84; do OFA_GEMM_TILE=$t OFA_SWEEP_CHECK=1 timeout 300 python tools/gemm_tile_sweep.py 1; done) > gpurun_out/q_sweep_p8v2.txt 2>&1
grep -v amdgpu gpurun_out/q_sweep_p8v2.txt
