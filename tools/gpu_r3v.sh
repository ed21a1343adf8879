#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3v; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "batchnorm" > $O/t.log 2>&1; echo "rc=$?"; tail -2 $O/t.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/st -o p -- python $R/bench.py --workload cfg2b --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/run.log 2>&1
python $R/tools/prof_summary.py /tmp/st/p_results.db 24 70 > $O/stats_cfg2b.txt 2>&1
head -1 $O/stats_cfg2b.txt; grep "bn_" $O/stats_cfg2b.txt | cut -c1-120
cd $R; python tools/bn_bench.py 2>&1 | grep rows
