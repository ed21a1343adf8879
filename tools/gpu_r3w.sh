#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/a -o p -- python $R/bench.py --workload cfg2b --steps 40 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/a.log 2>&1
python $R/tools/prof_summary.py /tmp/a/p_results.db 54 200 > $O/graph54.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/b -o p -- python $R/bench.py --workload cfg2b --steps 6 --warmup 2 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/b.log 2>&1
python $R/tools/prof_summary.py /tmp/b/p_results.db 8 200 > $O/eager8.txt 2>&1
for f in graph54 eager8; do echo $f; head -1 $O/$f.txt; grep -E "copyBuffer|FillFunctor<c10::BFloat16|copy_batched|CUDAFunctor_add<c10::BFloat16>, std" $O/$f.txt | cut -c1-130; done
