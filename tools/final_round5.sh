#!/bin/bash
# The last lease of round 5 (what the remaining GPU minutes allowed): cfg-2b sanity on the shipped library, the cfg-2 kernel trace, the cfg-2 bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/profiles; mkdir -p $O
cd $R
timeout 110 python bench.py --workload cfg2b --steps 10 --warmup 3 --no-cpu-baseline > $O/round5_bench_cfg2b.json 2> $O/bench_cfg2b_run.log; echo "cfg2b rc=$?"; tail -c 400 $O/round5_bench_cfg2b.json
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/r5f -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r5f/p_results.db 24 70 --json $O/round5_rocprof_kernel_stats.json > $O/round5_rocprof_kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r5f/p_results.db > $O/round5_rocprof_by_grid.txt 2>&1
python $R/tools/prof_last_step.py /tmp/r5f/p_results.db > $O/round5_last_step_sequence.txt 2>&1
cd $R
cp $O/round5_rocprof_kernel_stats.json $R/profiles/
timeout 200 python bench.py > $O/round5_bench.json 2> $O/bench_run.log; echo "bench rc=$?"
tail -c 1500 $O/round5_bench.json
