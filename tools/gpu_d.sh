#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r2stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/d_stats_run.log 2>&1
# steps executed: 4 batches x 3 setup (2 eager + capture/replay) + 2 + 10
python $R/tools/prof_summary.py /tmp/r2stats/p_results.db 1 70 > $O/d_kernel_stats_raw.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r2stats/p_results.db > $O/d_by_grid.txt 2>&1
tail -2 $O/d_stats_run.log
head -50 $O/d_kernel_stats_raw.txt
