#!/bin/bash
# round 3, call A: full GPU suite (all failures shown), smoke, default bench, kernel-trace of the bench command
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -30
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.log; tail -c 1500 $O/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r3stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $R/$O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r3stats/p_results.db 24 80 --json $R/$O/kernel_stats.json > $R/$O/kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r3stats/p_results.db > $R/$O/by_grid.txt 2>&1
head -5 $R/$O/kernel_stats.txt
