#!/bin/bash
# round 3, call O: copy_batched test + kernel traces of the three workloads
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3o; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "copy_batched or maxpool or conv2d or bias" > $O/t.log 2>&1; echo "rc=$?"; tail -2 $O/t.log; grep -E "^FAILED|^E  " $O/t.log | head
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg2b cfg4; do
  rocprofv3 --kernel-trace --stats -d /tmp/st_$w -o p -- python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_${w}_run.log 2>&1
  python $R/tools/prof_summary.py /tmp/st_$w/p_results.db 24 70 > $O/stats_$w.txt 2>&1
  head -42 $O/stats_$w.txt | cut -c1-150
done
