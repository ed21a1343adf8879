#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_sbias_gpu.py tests/test_configs_gpu.py -q -m gpu -x > $O/t.log 2>&1; rc=$?; echo "gate rc=$rc"; tail -3 $O/t.log; grep -E "^FAILED|^E  " $O/t.log | head
if [ $rc -ne 0 ]; then exit 1; fi
timeout 600 python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/bench_cfg4.json 2> $O/bench_cfg4.log
python -c "
import json;d=json.load(open('$O/bench_cfg4.json'));print('cfg4', round(d['ms_per_step'],3), round(d['value']))"
