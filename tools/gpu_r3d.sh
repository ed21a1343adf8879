#!/bin/bash
# round 3, call D: dsum v2 + planned table gradient: tests, microbench, cfg-2b / cfg-4 / cfg-2 bench, cfg-2b profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_attn_sbias_gpu.py -q > $O/t_sbias.log 2>&1; echo "sbias rc=$?"; tail -4 $O/t_sbias.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_attn_sbias_gpu.py > $O/t_all.log 2>&1; echo "all rc=$?"
grep -E "passed|failed|error" $O/t_all.log | tail -3; grep -E "^FAILED|^ERROR" $O/t_all.log | head -30
for w in cfg2b cfg4 dec cross; do timeout 300 python tools/attn_sbias_bench.py $w >> $O/attn_sbias_bench.txt 2>&1; done
grep -v amdgpu.ids $O/attn_sbias_bench.txt
for w in cfg2b cfg4 cfg2; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.log; python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', d['ms_per_step'], d['value'], d['config']['ragged_row_packing'], d['roofline'].get('frac'), d['roofline'].get('frac_raw_events'))" || tail -5 $O/bench_$w.log
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r3d2b -o p -- python $R/bench.py --workload cfg2b --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $R/$O/stats2b_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r3d2b/p_results.db 24 100 --json $R/$O/cfg2b_kernel_stats.json > $R/$O/cfg2b_kernel_stats.txt 2>&1
head -75 $R/$O/cfg2b_kernel_stats.txt
