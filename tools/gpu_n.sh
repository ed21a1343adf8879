#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r2b -o p -- python $R/bench.py --workload cfg2b --steps 6 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/n_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r2b/p_results.db 20 45 > $O/n_cfg2b_kernel_stats.txt 2>&1
head -48 $O/n_cfg2b_kernel_stats.txt
