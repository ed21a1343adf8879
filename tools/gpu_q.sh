#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r2stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $R/gpurun_out/q_stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r2stats/p_results.db 24 45 > $R/gpurun_out/q_kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r2stats/p_results.db > $R/gpurun_out/q_by_grid.txt 2>&1
head -46 $R/gpurun_out/q_kernel_stats.txt | cut -c1-150
grep -i attn $R/gpurun_out/q_by_grid.txt | head -12 | cut -c1-150
