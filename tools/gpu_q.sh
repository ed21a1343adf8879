#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export OFA_SWEEP_SHAPES="TN,2304,768,13312;TN,768,768,13312;TN,3072,768,13312;TN,768,3072,13312;TN,2304,768,2048;TN,3072,768,2048"
for bs in 0 1; do
OFA_GEMM_BIGSPLIT=$bs rocprofv3 --kernel-trace --stats -d /tmp/bs$bs -o p -- python $R/tools/gemm_tile_sweep.py 1 > $R/gpurun_out/q_bs$bs.log 2>&1
echo "== BIGSPLIT=$bs"; grep -v amdgpu $R/gpurun_out/q_bs$bs.log | grep tile
python $R/tools/prof_by_grid.py /tmp/bs$bs/p_results.db | grep -i "gemm\|splitk" | head -14 | cut -c1-170
done
