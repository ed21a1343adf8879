#!/bin/bash
# round 3, call B: the positional attention kernels -- unit tests first (under timeout: new kernels), then the model-level goldens, microbench, cfg-2b / cfg-4 bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_attn_pos_gpu.py -q -x > $O/t_attn_pos.log 2>&1; echo "attn_pos rc=$?"; tail -5 $O/t_attn_pos.log
timeout 900 python -m pytest tests/test_packing_gpu.py tests/test_model_gpu.py tests/test_fp16_gpu.py tests/test_configs_gpu.py "tests/test_bench_parity_gpu.py::test_large_with_default_image_adaptor_vs_oracle" -q -m gpu > $O/t_models.log 2>&1; echo "models rc=$?"
grep -E "passed|failed|error" $O/t_models.log | tail -3; grep -E "^FAILED|^ERROR" $O/t_models.log | head -30
timeout 300 python tools/attn_pos_bench.py cfg2b > $O/attn_pos_bench.txt 2>&1
OFA_ATTN_POS_DQ_W2=1 timeout 300 python tools/attn_pos_bench.py cfg2b >> $O/attn_pos_bench.txt 2>&1
timeout 300 python tools/attn_pos_bench.py cfg4 >> $O/attn_pos_bench.txt 2>&1
OFA_ATTN_POS_DQ_W2=1 timeout 300 python tools/attn_pos_bench.py cfg4 >> $O/attn_pos_bench.txt 2>&1
cat $O/attn_pos_bench.txt
for w in cfg2b cfg4; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.log; python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', d['ms_per_step'], d['value'], d['config']['ragged_row_packing'])" || tail -5 $O/bench_$w.log
done
timeout 600 python bench.py --workload cfg2b --no-pack --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2b_nopack.json 2> $O/bench_cfg2b_nopack.log; python -c "
import json;d=json.load(open('$O/bench_cfg2b_nopack.json'));print('cfg2b nopack', d['ms_per_step'], d['value'])"
