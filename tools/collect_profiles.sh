#!/bin/bash
# Round-1 evidence: kernel-trace stats of the default bench command, HBM PMC passes, microbenchmarks.  Run on the GPU box.
set -x
R=/root/repo; O=$R/gpurun_out/profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r1stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r1stats/p_results.db 15 60 > $O/round1_rocprof_kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r1stats/p_results.db > $O/round1_rocprof_by_grid.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r1fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r1write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/write_run.log 2>&1
python $R/tools/pmc_traffic.py /tmp/r1fetch/p_results.db /tmp/r1write/p_results.db $O/round1_pmc_traffic.json > $O/round1_pmc_traffic.txt 2>&1
cd $R
python tools/gemm_bench.py > $O/round1_gemm_microbench.txt 2>&1
python tools/attn_bench.py >> $O/round1_gemm_microbench.txt 2>&1
python tools/decode_bench.py > $O/round1_decode_microbench.txt 2>&1
python bench.py > $O/round1_bench.json 2> $O/bench_run.log
tail -c 600 $O/round1_bench.json
