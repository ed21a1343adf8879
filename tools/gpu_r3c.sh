#!/bin/bash
# round 3, call C: shared position bias -- kernel tests (under timeout), the model-level goldens, microbench, cfg-2b / cfg-4 bench + kernel trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_attn_sbias_gpu.py -q > $O/t_sbias.log 2>&1; echo "sbias rc=$?"; tail -4 $O/t_sbias.log
timeout 1200 python -m pytest tests/test_packing_gpu.py tests/test_model_gpu.py tests/test_fp16_gpu.py tests/test_configs_gpu.py tests/test_bench_parity_gpu.py tests/test_kernels_gpu.py -q -m gpu > $O/t_models.log 2>&1; echo "models rc=$?"
grep -E "passed|failed|error" $O/t_models.log | tail -3; grep -E "^FAILED|^ERROR" $O/t_models.log | head -30
for w in cfg2b cfg4 dec cross; do timeout 300 python tools/attn_sbias_bench.py $w >> $O/attn_sbias_bench.txt 2>&1; done
cat $O/attn_sbias_bench.txt
for w in cfg2b cfg4; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.log; python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', d['ms_per_step'], d['value'], d['config']['ragged_row_packing'])" || tail -5 $O/bench_$w.log
done
timeout 600 python bench.py --workload cfg2b --no-pack --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2b_nopack.json 2> $O/bench_cfg2b_nopack.log; python -c "
import json;d=json.load(open('$O/bench_cfg2b_nopack.json'));print('cfg2b nopack', d['ms_per_step'], d['value'])"
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r3c2b -o p -- python $R/bench.py --workload cfg2b --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $R/$O/stats2b_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r3c2b/p_results.db 24 90 --json $R/$O/cfg2b_kernel_stats.json > $R/$O/cfg2b_kernel_stats.txt 2>&1
head -60 $R/$O/cfg2b_kernel_stats.txt
