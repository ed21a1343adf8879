#!/bin/bash
# round 3, call H: gate for the Adam / split-K planner / BatchNorm fold changes, glue attribution, quick benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_trainstep_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/t_gate.log 2>&1; echo "gate rc=$?"; tail -3 $O/t_gate.log; grep -E "^FAILED|^E  " $O/t_gate.log | head
timeout 600 python tools/native_glue_trace.py cfg2b 2>&1 | grep -v amdgpu.ids > $O/glue_cfg2b.txt; head -75 $O/glue_cfg2b.txt | cut -c1-260
timeout 600 python tools/native_glue_trace.py cfg2 2>&1 | grep -v amdgpu.ids > $O/glue_cfg2.txt; head -45 $O/glue_cfg2.txt | cut -c1-260
for w in cfg2 cfg2b cfg4; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/bench_$w.json 2> $O/bench_$w.log
  python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', round(d['ms_per_step'],3), round(d['value']))"
done
