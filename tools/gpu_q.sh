#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r2b -o p -- python $R/bench.py --workload cfg2b --steps 6 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats2b_run.log 2>&1
grep -c . $O/stats2b_run.log; tail -1 $O/stats2b_run.log | cut -c1-200
python $R/tools/prof_summary.py /tmp/r2b/p_results.db 20 60 > $O/cfg2b_kernel_stats.txt 2>&1
head -50 $O/cfg2b_kernel_stats.txt | cut -c1-160
