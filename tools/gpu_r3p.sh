#!/bin/bash
# round 3, call P: outer (video) rel-pos slots: gate + video model tests + benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3p; mkdir -p $O
timeout 1500 python -m pytest tests/test_attn_sbias_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_bench_parity_gpu.py -q -m gpu -x > $O/t_gate.log 2>&1; rc=$?; echo "gate rc=$rc"; tail -3 $O/t_gate.log; grep -E "^FAILED|^E  " $O/t_gate.log | head -20
if [ $rc -ne 0 ]; then exit 1; fi
for w in cfg2b cfg4; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/bench_$w.json 2> $O/bench_$w.log
  python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', round(d['ms_per_step'],3), round(d['value']))"
done
