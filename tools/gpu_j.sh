#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -q -m gpu --timeout 1200 -x > gpurun_out/j_all_gpu_tests.log 2>&1
tail -3 gpurun_out/j_all_gpu_tests.log
for w in cfg2b cfg4; do
python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/j_bench_$w.json 2> gpurun_out/j_bench_$w.err
python - <<PY
import json
try:
    o=json.loads(open("gpurun_out/j_bench_$w.json").read().strip().splitlines()[-1]); r=o["roofline"]
    print("$w", round(o["ms_per_step"],3), round(o["value"]), round(r["step_frac"],4), round(r.get("frac"),4), round(r.get("gemm_ms_per_step",0),3), o["config"]["step_mode"])
except Exception as e:
    print("$w ERR", e); print(open("gpurun_out/j_bench_$w.err").read()[-1500:])
PY
done
