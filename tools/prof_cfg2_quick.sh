R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof5a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r5a -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r5a/p_results.db 24 90 --json $O/kernel_stats.json > $O/kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r5a/p_results.db > $O/by_grid.txt 2>&1
tail -3 $O/stats_run.log | cut -c1-300
python $R/tools/prof_last_step.py /tmp/r5a/p_results.db > $O/last_step.txt 2>&1
