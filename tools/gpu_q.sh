#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_trainstep_gpu.py tests/test_model_gpu.py tests/test_packing_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== bench grouped"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
echo "== bench ungrouped"; OFA_WGRAD_GROUP=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r2stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r2stats/p_results.db 24 40 > $O/kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r2stats/p_results.db > $O/by_grid.txt 2>&1
head -12 $O/kernel_stats.txt | cut -c1-150
grep -n "group\|fold\|Lb0ELb0ELb0ELb1" $O/by_grid.txt | cut -c1-170
