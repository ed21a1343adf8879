#!/bin/bash
# round 3, call E: cross-KV batching + dsum v3: full GPU suite, microbench, three benches, cfg-2 + cfg-2b kernel traces
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"
grep -E "passed|failed|error" $O/t_all.log | tail -3; grep -E "^FAILED|^ERROR" $O/t_all.log | head -30
for w in cfg2b cfg4 dec cross; do timeout 300 python tools/attn_sbias_bench.py $w >> $O/attn_sbias_bench.txt 2>&1; done
grep -v amdgpu.ids $O/attn_sbias_bench.txt
for w in cfg2 cfg2b cfg4; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.log; python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', d['ms_per_step'], d['value'], d['config']['ragged_row_packing'], d['roofline'].get('frac'), d['roofline'].get('frac_raw_events'))" || tail -5 $O/bench_$w.log
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r3e2 -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $R/$O/stats2_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r3e2/p_results.db 24 60 --json $R/$O/cfg2_kernel_stats.json > $R/$O/cfg2_kernel_stats.txt 2>&1
head -40 $R/$O/cfg2_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r3e2b -o p -- python $R/bench.py --workload cfg2b --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $R/$O/stats2b_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r3e2b/p_results.db 24 100 --json $R/$O/cfg2b_kernel_stats.json > $R/$O/cfg2b_kernel_stats.txt 2>&1
head -30 $R/$O/cfg2b_kernel_stats.txt
