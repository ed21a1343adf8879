#!/bin/bash
# round 3, call J: gate for the vectorised bias build, then a kernel trace of the cfg-2b / cfg-4 steps
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_sbias_gpu.py tests/test_packing_gpu.py -q -m gpu -x > $O/t_sbias.log 2>&1; rc=$?; echo "sbias rc=$rc"; tail -3 $O/t_sbias.log; grep -E "^FAILED|^E  " $O/t_sbias.log | head -20
if [ $rc -ne 0 ]; then exit 1; fi
for w in cfg2b cfg4; do python tools/attn_sbias_bench.py $w 2>&1 | grep -E "shared"; done
cd /tmp && export TMPDIR=/tmp
for w in cfg2b cfg4; do
  rocprofv3 --kernel-trace --stats -d /tmp/st_$w -o p -- python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_${w}_run.log 2>&1
  python $R/tools/prof_summary.py /tmp/st_$w/p_results.db 24 60 > $O/stats_$w.txt 2>&1
  head -48 $O/stats_$w.txt | cut -c1-170
done
cd $R
timeout 600 python tools/native_glue_trace.py cfg2b 2>&1 | grep -v amdgpu.ids > $O/glue_cfg2b.txt; head -40 $O/glue_cfg2b.txt | cut -c1-230
