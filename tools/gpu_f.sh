#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_packing_gpu.py -q -m gpu --timeout 900 > gpurun_out/f_pack_tests.log 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " gpurun_out/f_pack_tests.log | head -20
python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 900 -k "attn or attention or gemm" > gpurun_out/f_k_tests.log 2>&1
tail -3 gpurun_out/f_k_tests.log
for mode in "" "--no-pack"; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline $mode > gpurun_out/f_bench$mode.json 2> gpurun_out/f_bench$mode.err
python - <<PY
import json
o=json.loads(open("gpurun_out/f_bench$mode.json").read().strip().splitlines()[-1]); r=o["roofline"]
print("$mode", round(o["ms_per_step"],3), round(o["value"]), round(r["step_frac"],4), round(r.get("frac"),4), round(r.get("gemm_ms_per_step"),3), o["config"]["ragged_row_packing"])
PY
done
